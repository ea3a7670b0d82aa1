"""How much would ray re-ordering buy the BVH walk?  (round 2's verdict: "no ray reordering")
Bounce-like rays on the C4 / C5 stand-ins - origins on random triangles, cosine-distributed directions about the normal - are traced with
apt_intersect (k_extend<bvh>, one flat queue) in three orders: as generated (incoherent), sorted by direction octant only, and sorted by
(octant, 30-bit Morton code of the origin) - the best a sort pass in front of the extend stage could deliver.
    rocprofv3 --kernel-trace -d gpurun_out/coh -o coh -- python tools/gpu_coherence_probe.py c4 ; python tools/gpu_coherence_probe.py --report gpurun_out/coh
"""
import glob, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def morton3(q):
    def spread(x):
        x = x.astype(np.uint64) & 0x3ff
        x = (x | (x << 16)) & 0x30000ff; x = (x | (x << 8)) & 0x300f00f; x = (x | (x << 4)) & 0x30c30c3; x = (x | (x << 2)) & 0x9249249
        return x
    return spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)


def main(cfg):
    from adapt_amd import synth
    from adapt_amd.renderer import Renderer
    tup = synth.three_bunnies() if cfg == "c4" else synth.bunny_field()
    prims = tup[1]["primitives"]; ng = tup[1]["n_g"]
    rs = np.random.RandomState(3)
    n = 4_000_000
    k = rs.randint(prims.shape[0], size=n)
    u, v = rs.uniform(size=n), rs.uniform(size=n); flip = u + v > 1; u[flip], v[flip] = 1 - u[flip], 1 - v[flip]
    p = prims[k, 0] * (1 - u - v)[:, None] + prims[k, 1] * u[:, None] + prims[k, 2] * v[:, None]
    nrm = ng[k]
    a = np.cross(nrm, np.where(np.abs(nrm[:, :1]) < 0.9, [[1., 0., 0.]], [[0., 1., 0.]])); a /= np.linalg.norm(a, axis=1, keepdims=True)
    b = np.cross(nrm, a)
    r1, r2 = rs.uniform(size=n), rs.uniform(size=n)
    d = np.sqrt(r1)[:, None] * (np.cos(2 * np.pi * r2)[:, None] * a + np.sin(2 * np.pi * r2)[:, None] * b) + np.sqrt(1 - r1)[:, None] * nrm
    o = (p + 1e-3 * nrm).astype(np.float32); d = d.astype(np.float32)
    octant = (d[:, 0] < 0) * 4 + (d[:, 1] < 0) * 2 + (d[:, 2] < 0) * 1
    lo, hi = o.min(0), o.max(0)
    code = morton3(((o - lo) / (hi - lo + 1e-9) * 1023).astype(np.int64))
    orders = {"as generated": np.arange(n), "by octant": np.argsort(octant, kind="stable"), "by octant + origin": np.lexsort((code, octant)), "by origin": np.argsort(code, kind="stable")}
    r = Renderer(*tup, width=64, height=64, spp_per_batch=1024)
    ref = None
    for name, idx in orders.items():
        prim, t, _ = r.intersect(o[idx], d[idx])
        back = np.empty(n, np.int64); back[idx] = np.arange(n)
        if ref is None: ref = (prim, t)
        assert np.array_equal(prim[back], ref[0]) and np.array_equal(t[back], ref[1])
        print("traced", name, flush=True)
    r.close()


def report(d):
    import sqlite3
    f = sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True))[-1]
    db = sqlite3.connect(f)
    rows = list(db.execute("select d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
                           "where s.kernel_name like '%k_extend%' order by d.start"))
    for name, (a, b) in zip(["as generated", "by octant", "by octant + origin", "by origin"], rows[-4:]):
        ms = (b - a) * 1e-6
        print(f"{name:22s} {ms:8.3f} ms   {4.0 / ms:6.2f} G rays/s")


if __name__ == "__main__":
    if sys.argv[1] == "--report": report(sys.argv[2])
    else: main(sys.argv[1])
