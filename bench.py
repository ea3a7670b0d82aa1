#!/usr/bin/env python3
"""bench.py — Msamples/s of the HIP wavefront path tracer on BASELINE.json's headline config.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c1..c5|v1..v3] [--scaling weak|strong]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

One "step" = one complete render of the workload: every owned pixel receives its samples through the
generate/extend/shade/shadow/finalize stages (+ the tile all-gather when N > 1).  N = 1 workload = BASELINE
configs[1]: cornell box, 512x512, 1024 spp, 8 bounces.  For N > 1 the film is sharded in interleaved column
bands; `--scaling weak` (default, except c4 / c5) scales the sample count by N so that every GPU does the N = 1 amount of work,
`--scaling strong` keeps film and samples fixed (what BASELINE configs 4 / 5 describe: one image split over 4 / 8
GPUs).  The only collective is the all_gather of the tile framebuffers over RCCL.

Prints ONE JSON line on rank 0 with the contract's fields plus
  roofline     : per stage kernel (extend / shade / shadow) the algorithmic bytes per launch / mean launch time from
                 HIP events on the stream the kernel is launched on -> GB/s against the 8 TB/s HBM3E peak, and next to
                 it the VALU roofline of the same kernel (instructions issued per SIMD against the measured shader
                 clock); the headline fields are those of the dominant kernel
  cpu_baseline : the CPU oracle (C port of the reference path) on a bounded sample of the same workload
  parity       : HIP vs that CPU render of the same pixels/samples/seed (per-pixel L2 -> relMSE, max abs)
"""
import argparse
import glob
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# film sharding for N > 1: interleaved column bands.  Measured on C2 (every rank of an 8-rank step rendered in turn on one GPU,
# tools/gpu_rank_probe.py): slowest / mean rank time 1.196 with 32-column bands (the lit centre of the image is dearer than its dark
# edges), 1.032 with 16, 1.016 with 4; the per-rank rate itself does not depend on the band width.
BAND_WIDTH = 4
HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
ROUND = 6                       # profiles/r0<ROUND>_<config>_counters.json: PMC figures of THIS round's kernels
STAGES = ("extend", "shade", "shadow")

CONFIGS = {
    # name: (scene dir, file, width, height, spp, max_bounce, label)
    "c1": ("cbox", "c2_cbox.xml", 256, 256, 64, 4, "cbox 256x256, 64 spp, 4 bounces (BASELINE configs[0])"),
    "c2": ("cbox", "c2_cbox.xml", 512, 512, 1024, 8, "cbox 512x512, 1024 spp, 8 bounces (BASELINE configs[1])"),
    "c3": ("csphere", "c3_balls_mono.xml", 512, 512, 1024, 16, "csphere balls-mono 512x512, 1024 spp, 16 bounces (BASELINE configs[2])"),
    # synthetic stand-ins (adapt_amd/synth.py): the reference does not ship these scenes' assets
    "c4": ("synth", "three-bunnies", 800, 800, 512, 8, "three-bunnies stand-in, 95 050 tris, 800x800, 512 spp, 8 bounces (BASELINE configs[3])"),
    "c5": ("synth", "bunny-field", 1280, 720, 2048, 16, "sports-car stand-in (bunny field), 285 134 tris, 1280x720, 2048 spp, 16 bounces (BASELINE configs[4])"),
    # volumetric path tracer (the reference's `--type vpt`): names starting with "v" render with VolumeRenderer
    "v1": ("vpt", "cbox_fog.xml", 512, 512, 256, 16, "vpt Cornell box (the reference's scenes/vpt/cbox.xml set-up: Lambertian box, quad light, fog cube = null surface + H-G medium); "
                                                    "512x512, 256 spp, 16 bounces, 1 light sample per vertex, volumetric tracer"),
    "v3": ("test", "volgrid_a.xml", 512, 512, 128, 8, "volgrid_a: Cornell box with a 20x16x12 RGB grid volume (delta / ratio tracking), mirror ball, quad + point light, "
                                                      "2 light samples per vertex; 512x512, 128 spp, 8 bounces, volumetric tracer"),
    "v2": ("test", "media_a.xml", 512, 512, 256, 8, "media_a: Cornell box, fog cube behind a null surface, scattering glass ball, thin multi-H-G world medium, "
                                                    "2 light samples per vertex; 512x512, 256 spp, 8 bounces, volumetric tracer (all-models kernel)"),
}


def load_scene(sdir, sfile):
    if sdir == "synth":
        from adapt_amd.synth import SYNTH_SCENES
        return SYNTH_SCENES[sfile]()
    from adapt_amd import scene_parsing
    cwd = os.getcwd()
    os.chdir(ROOT)                  # texture / .vol paths inside scene files are relative to the repository root
    try:
        return scene_parsing(os.path.join(ROOT, "scenes", sdir), sfile)
    finally:
        os.chdir(cwd)


def csrc_sha256():
    """Identity of the kernels: the PMC figures under profiles/ are only attached to a run of exactly these sources."""
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "adapt_amd", "csrc", "*"))):
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()


def kernel_bytes(st):
    """Algorithmic HBM bytes per stage from the path statistics (DESIGN.md 'bytes each stage moves'):
    4-byte SoA lanes; ray = o,d 24 B; hit = t,prim,u,v 16 B; state = throughput,id,meta,pdf 24 B;
    shadow entry = o,d,tmax,contribution,id 44 B; radiance L = 12 B (24 B per read-modify-write)."""
    n_s, n_e, n_sh, n_lit = st["n_samples"], st["n_extend"], st["n_shadow_traced"], st["n_lit"]
    n_cont = n_e - n_s                       # queue entries written by shade for the next bounce
    kb = {
        "generate": 60 * n_s,                # ray 24 + state 24 + zeroed L 12
        "extend": 40 * n_e,                  # read ray 24, write hit 16
        "shade": 64 * n_e + 48 * n_cont + 44 * n_sh,     # read ray+hit+state, write next ray+state, write shadow entries
        "shadow": 44 * n_sh + 24 * n_lit,    # read entry, RMW radiance of unoccluded ones
        "finalize": 12 * n_s,                # read L (framebuffer RMW is 24 B per pixel per batch: negligible)
    }
    if st.get("launches", {}).get("shadow", 1) == 0 and n_sh > 0:
        # rays traced in place (shade_stage.hpp k_shade_traced): the shade kernel does the shadow stage's work as well, so the UNIT of work
        # it is credited with is SURVEY 8(d)'s for both stages - 88 B per shadow ray + 24 B per unoccluded one on top of its own - although
        # the entries never travel through HBM (what it really moves is the `traffic` figure: 64 + 12 B read, 48 + 12 B written per entry)
        kb["shade"] += kb["shadow"]
        kb["shadow"] = 0
    if st.get("launches", {}).get("extend", 1) == 0 and n_e > 0:
        # rays traced in place (shade_stage.hpp "rays traced in place"): k_generate sweeps the camera rays and the shade kernel its continuation ray,
        # there is no extend launch - by the same rule the two kernels are credited with the extend stage's 40 B per ray they trace
        kb["generate"] += 40 * n_s
        kb["shade"] += 40 * n_cont
        kb["extend"] = 0
    return kb


def stage_units(st):
    return {"extend": st["n_extend"], "shade": st["n_extend"], "shadow": st["n_shadow_traced"]}


def region_roofline(stats, counters, n_simd, sclk_mhz):
    """Per-kernel figures of one measured region (HIP-event times from apt_get_stats)."""
    kb, kms, units = kernel_bytes(stats), stats["kernel_ms"], stage_units(stats)
    per = {}
    for k in kms:
        if stats["launches"][k] == 0 and kb[k] == 0:
            continue                                 # a stage this render does not launch (its fix-up time is in the bucket, nothing else)
        launches = max(1, stats["launches"][k])
        avg_ms = kms[k] / launches
        e = {"ms": round(kms[k], 3), "launches": int(stats["launches"][k]), "avg_launch_ms": round(avg_ms, 5), "alg_bytes": int(kb[k]),
             "bytes_per_launch": int(kb[k] / launches), "GB/s": round(kb[k] / (kms[k] * 1e-3) / 1e9, 1) if kms[k] > 0 else 0.0}
        e["frac"] = round(e["GB/s"] / HBM_PEAK_GBS, 5)
        if counters and k in counters["kernels"] and k in units and kms[k] > 0:
            c = counters["kernels"][k]
            e["traffic"] = int(c["bytes_per_unit"] * units[k] / launches)            # HBM-side bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE)
            e["traffic_over_algorithmic"] = round(c["bytes_per_unit"] * units[k] / max(1, kb[k]), 2)
            if "valu_insts_per_unit" in c and sclk_mhz:
                insts = c["valu_insts_per_unit"] * units[k]                           # wave-instructions, 4 cycles each on a SIMD16
                peak = n_simd * sclk_mhz * 1e6 / 4.0
                e["valu"] = {"Ginst/s": round(insts / (kms[k] * 1e-3) / 1e9, 2), "peak_Ginst/s": round(peak / 1e9, 2),
                             "busy_frac": round(insts / (kms[k] * 1e-3) / peak, 4), "insts_per_64_units": round(64 * c["valu_insts_per_unit"], 1)}
        per[k] = e
    return per


def pick_dominant(per):
    """Largest summed time among the stage kernels; a lead of less than 15 % does not count (the order extend > shade > shadow decides),
    so that two kernels at 34.9 % / 34.7 % of the time - or C3's shade and shadow, 6 % apart - cannot flip the headline from run to run."""
    best = max(per[k]["ms"] for k in STAGES if k in per)
    for k in STAGES:
        if k in per and per[k]["ms"] >= 0.85 * best:
            return k
    return STAGES[0]


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start N ranks of this very command line under torch.distributed.run."""
    import socket
    import subprocess
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL's device-buffer exchange between the ranks needs it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def host_cpu_report(omp_threads):
    """What the CPU leg can run on: OpenMP's idea of the machine, the affinity mask of this process, the cgroup CPU quota and the CPU model.
    (Round 5's line said "128 threads" where the 8-thread and the all-thread rates were equal: the box's container is granted a fraction
    of the host, which omp_get_max_threads() does not see.)  threads_used = min of the three, at least 1."""
    rep = {"omp_max_threads": int(omp_threads), "os_cpu_count": os.cpu_count()}
    try:
        rep["sched_affinity"] = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        rep["sched_affinity"] = None
    quota = None
    try:                                            # cgroup v2: "max 100000" or "<quota> <period>"
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        rep["cgroup_cpu_max"] = f"{q} {per}"
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:                                        # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            rep["cgroup_cpu_max"] = f"{q} {per}"
            if q > 0 and per > 0:
                quota = q / per
        except (OSError, ValueError):
            rep["cgroup_cpu_max"] = None
    rep["cgroup_quota_cpus"] = None if quota is None else round(quota, 2)
    model = None
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                model = line.split(":", 1)[1].strip(); break
    except OSError:
        pass
    rep["cpu_model"] = model
    try:
        rep["loadavg_1min"] = round(os.getloadavg()[0], 2)
    except OSError:
        rep["loadavg_1min"] = None
    limits = [int(omp_threads)] + ([rep["sched_affinity"]] if rep["sched_affinity"] else []) + ([max(1, int(quota + 0.5))] if quota else [])
    rep["threads_used"] = max(1, min(limits))
    return rep


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"], help="N > 1: weak = samples per step x N (per-GPU work fixed), strong = film and samples fixed; "
                    "default: strong for c4 / c5 (BASELINE describes them as ONE image tiled over 4 / 8 GPUs), weak otherwise")
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel per step")
    ap.add_argument("--spp-per-batch", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the bounded oracle sample")
    ap.add_argument("--lanes", type=int, default=0, help="concurrent render lanes (batch pipelines on separate HIP streams); 0 = library default (3)")
    ap.add_argument("--no-exclusive-pass", action="store_true", help="skip the extra untimed one-lane pass that measures every kernel alone")
    ap.add_argument("--no-profile", action="store_true", help="skip the repeat of the timed region with per-launch HIP events (the timed region itself never carries them) and the exclusive pass")
    ap.add_argument("--strict-profiles", action="store_true", help="fail if profiles/r0N_<config>_counters.json is missing or was taken on other kernels")
    ap.add_argument("--dump-image", default="", help="rank 0 writes the gathered (W, H, 3) accumulation of the timed region to this .npy file")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only for dry runs)")
    ap.add_argument("--single-device", action="store_true", help="dry run: every rank renders on cuda:0 (use with --backend gloo on a one-GPU box)")
    args = ap.parse_args()
    if args.dump_image:
        args.dump_image = os.path.abspath(args.dump_image)
    os.chdir(ROOT)                  # texture / .vol paths inside scene files are relative to the repository root (rocprofv3 runs from /tmp)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        # plain `python bench.py --gpus N`: no launcher has set the rendezvous variables, so this process becomes the launcher - it
        # re-executes itself under torch.distributed.run (one process per GPU, 127.0.0.1 rendezvous on a free port) and hands back
        # the ranks' exit code; rank 0 of that run prints the JSON line on this process's stdout
        sys.exit(self_launch(args.gpus))

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        args.gpus = world               # the launcher's world is authoritative (torch.distributed.run --nproc-per-node)
    if not torch.cuda.is_available():
        sys.exit("bench.py: no HIP device visible; the render path has no CPU fallback")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    from adapt_amd import _lib
    from adapt_amd.renderer import Renderer, VolumeRenderer
    from adapt_amd.tiles import assemble, device_tile, gather_tiles_device
    volumetric = args.config.startswith("v")
    if volumetric:
        Renderer = VolumeRenderer

    sdir, sfile, W, H, spp, bounces, label = CONFIGS[args.config]
    if args.lanes > 0:
        os.environ["APT_LANES"] = str(args.lanes)            # read by libadapt_mi when a renderer is created
    lanes = int(os.environ.get("APT_LANES", "3"))
    if args.spp > 0:
        spp = args.spp
    if args.scaling is None:
        args.scaling = "strong" if args.config in ("c4", "c5") else "weak"
    spp_step = spp * world if args.scaling == "weak" else spp       # weak: per-GPU samples stay at the N = 1 amount
    parsed = load_scene(sdir, sfile)
    rdr = Renderer(*parsed, width=W, height=H, max_bounce=bounces, device=local_rank, rank=rank, world_size=world,
                   band_width=BAND_WIDTH, profile=False, spp_per_batch=args.spp_per_batch)      # the timed region carries no per-launch events (below)
    info = rdr.info()

    gathered = [None]
    gather_ms = []

    def step():
        rdr.render(n_spp=spp_step)
        if world > 1:
            rdr.synchronize()                            # the tile must be complete before the collective reads it
            t0 = time.perf_counter()
            if args.backend == "nccl":
                gathered[0] = gather_tiles_device(device_tile(rdr), rdr.plan, world)     # all_gather over RCCL, result stays on the device
                torch.cuda.synchronize()
            else:
                gathered[0] = gather_tiles_device(torch.from_numpy(rdr.tile_accum()), rdr.plan, world)
            gather_ms.append((time.perf_counter() - t0) * 1e3)
        else:
            rdr.synchronize()

    def fence():
        if dist is not None:
            dist.barrier()
        rdr.synchronize()
        torch.cuda.synchronize()

    # one-off initialisation that is not a step: load every kernel of the pipeline (1 spp) and, for N > 1, create the RCCL
    # communicator with a first gather - so that `--warmup 0` does not time module loading or communicator set-up
    rdr.render(n_spp=1)
    if world > 1:
        rdr.synchronize()
        gather_tiles_device(device_tile(rdr) if args.backend == "nccl" else torch.from_numpy(rdr.tile_accum()), rdr.plan, world)
    rdr.synchronize()
    for _ in range(args.warmup):
        step()
    rdr.clear()                              # zero accumulation + statistics + event timers: the timed region starts clean
    gather_ms.clear()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt_local = time.perf_counter() - t0
    dt = dt_local
    st = rdr.stats()
    per_rank = None
    if dist is not None:
        dev = f"cuda:{local_rank}" if args.backend == "nccl" else "cpu"
        mine = torch.tensor([dt_local, st["render_ms"] / max(1, args.steps), float(np.mean(gather_ms)) if gather_ms else 0.0, float(st["n_samples"]), float(local_rank)], dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        allr = torch.stack(allr).cpu().numpy()
        dt = float(allr[:, 0].max())                                    # the job is as slow as its slowest rank
        per_rank = {"wall_s": [round(float(x), 5) for x in allr[:, 0]], "render_ms_per_step": [round(float(x), 3) for x in allr[:, 1]],
                    "gather_ms_per_step": [round(float(x), 3) for x in allr[:, 2]], "samples": [int(x) for x in allr[:, 3]],
                    "collective": {"backend": "RCCL (torch.distributed 'nccl')" if args.backend == "nccl" else args.backend, "world_size": world,
                                   # what the process group itself reports (did the collective see N ranks, and which library carried it)
                                   "process_group": {"backend": str(dist.get_backend()), "world_size": int(dist.get_world_size()),
                                                     "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if args.backend == "nccl" else None,
                                                     "devices": sorted(set(int(x) for x in allr[:, 4]))},
                                   "op": "all_gather_into_tensor of the per-rank (n_cols, H, 3) float32 tiles, once per step",
                                   "tile_bytes_per_rank": int(rdr.n_cols * rdr.h * 12), "queue_MiB_per_rank": round(info["queue_bytes"] / 2 ** 20, 1)}}

    total_samples = W * H * spp_step * args.steps
    value = total_samples / dt / 1e6
    if args.dump_image and rank == 0:
        if world > 1:
            g = gathered[0]
            np.save(args.dump_image, assemble(rdr.plan, (g.cpu() if hasattr(g, "cpu") else g).numpy()))
        else:
            np.save(args.dump_image, rdr.tile_accum())

    # ---- what this process cannot measure from inside: HBM-side bytes and issued VALU instructions per queue unit, from the
    # rocprofv3 PMC passes of THIS round's kernels (tools/pmc_round.sh -> profiles/r0N_<config>_counters.json)
    counters, counters_note = None, None
    cfile = os.path.join(ROOT, "profiles", f"r{ROUND:02d}_{args.config}_counters.json")
    if os.path.exists(cfile):
        c = json.load(open(cfile))
        if c.get("csrc_sha256") == csrc_sha256():
            counters = c
        else:
            counters_note = f"{os.path.relpath(cfile, ROOT)} was recorded on other kernel sources (csrc hash differs): traffic / VALU figures withheld"
    else:
        counters_note = f"{os.path.relpath(cfile, ROOT)} not found: traffic / VALU figures withheld"
    if counters is None and args.strict_profiles:
        sys.exit("bench.py --strict-profiles: " + counters_note)

    sclk = None
    mhz = _lib.C.c_float(0)
    if _lib.load().apt_measure_sclk_mhz(local_rank, _lib.C.byref(mhz)) == 0:
        sclk = float(mhz.value)
    props = torch.cuda.get_device_properties(local_rank)
    n_simd = int(props.multi_processor_count) * 4

    # Per-launch HIP events are not free: one record per launch holds the next launch back until the previous one has retired and its
    # timestamp is written - 0.6 % of C2's rate (10 launches per batch), 1.5-3.6 % of C3 / C4 / C5's (130-200 launches per batch: a kernel
    # per material class and bounce).  So the TIMED region runs without them (`value` is the rate of the render as a user runs it), and the
    # same K steps are repeated with the events on (`roofline.timed_region`: per-kernel sums with the lanes overlapping, and what the
    # events cost: `event_overhead`).
    def second_renderer():
        """A renderer with per-launch events beside the timed one, at the batch size the timed one chose (its own choice is made with less free
        memory and could differ); when that does not fit - several ranks sharing one device - at the size the library then picks."""
        kw = dict(width=W, height=H, max_bounce=bounces, device=local_rank, rank=rank, world_size=world, band_width=BAND_WIDTH, profile=True)
        try:
            return Renderer(*parsed, spp_per_batch=info["spp_per_batch"], **kw)
        except _lib.AptError:
            return Renderer(*parsed, spp_per_batch=args.spp_per_batch, **kw)

    timed, overlap, prof = region_roofline(st, counters, n_simd, sclk), None, None
    if not args.no_profile:
        rp = second_renderer()
        rp.render(n_spp=1); rp.synchronize()
        rp.render(n_spp=spp_step); rp.synchronize(); rp.clear()
        tp = time.perf_counter()
        for _ in range(args.steps):
            rp.render(n_spp=spp_step); rp.synchronize()
        dt_prof = time.perf_counter() - tp
        stp = rp.stats()
        rp.close()
        timed = region_roofline(stp, counters, n_simd, sclk)
        overlap = round(sum(stp["kernel_ms"].values()) / stp["render_ms"], 3) if stp["render_ms"] > 0 else None
        prof = {"sum_kernel_ms": round(sum(stp["kernel_ms"].values()), 3), "render_ms": round(stp["render_ms"], 3),
                "Msamples/s": round(rdr.n_cols * rdr.h * spp_step * args.steps / dt_prof / 1e6, 1), "event_overhead": round(st["render_ms"] and stp["render_ms"] / st["render_ms"] - 1.0, 4)}
    # With more than one render lane, kernels of different batches run side by side on the GPU: a HIP-event bracket in the
    # timed region then measures a kernel that shares the machine (`overlap` = summed kernel time / wall time), which says
    # nothing about the kernel.  The per-kernel rooflines are therefore measured with the same events in an extra, untimed pass of
    # the same workload on ONE lane, where every kernel has the GPU to itself; the timed-region figures stay in `timed_region`.
    alone, source, one_lane_rate = timed, "repeat of the timed region with per-launch events (one render lane: kernels do not overlap)", None
    if lanes > 1 and not args.no_exclusive_pass and not args.no_profile:
        os.environ["APT_LANES"] = "1"
        r1 = second_renderer()
        os.environ["APT_LANES"] = str(lanes)
        n1 = max(1, min(spp_step, 256))
        r1.render(n_spp=n1); r1.synchronize(); r1.clear()
        r1.render(n_spp=n1); r1.synchronize()
        s1 = r1.stats()
        alone = region_roofline(s1, counters, n_simd, sclk)
        one_lane_rate = round(rdr.n_cols * rdr.h * n1 / (s1["render_ms"] * 1e-3) / 1e6, 1) if s1["render_ms"] > 0 else None
        source = f"exclusive pass after the timed region: same workload, {n1} spp, one render lane (kernels of concurrent lanes overlap in the timed region)"
        r1.close()
    elif lanes > 1:
        source = "repeat of the timed region with per-launch events, overlapping render lanes (exclusive pass disabled): per-kernel durations include co-scheduled kernels"
    dom = pick_dominant(alone)
    kb = kernel_bytes(st)
    # What binds the dominant kernel.  `frac_credited` prices its ALGORITHMIC bytes (SURVEY 8(d)'s per-unit figure - for a kernel that
    # absorbed other stages, theirs too) against the HBM peak; `frac_hbm_moved` prices the bytes the PMC counters saw cross the HBM side.
    # A kernel that moves fewer bytes than it is credited with (traffic / algorithmic < 1: the fused C1 / C2 kernel never writes its
    # rays, hits or light samples) is not waiting for HBM, and the headline fields then describe the resource it does wait for: VALU
    # issue (wave-instructions per second against n_simd x sclk / 4), with both HBM fractions kept beside it.
    d_ = alone[dom]
    moved = (d_["traffic"] / (d_["avg_launch_ms"] * 1e-3) / 1e9) if d_.get("traffic") and d_["avg_launch_ms"] > 0 else None
    hbm = {"achieved_credited_GB/s": d_["GB/s"], "frac_credited": d_["frac"], "achieved_moved_GB/s": round(moved, 1) if moved is not None else None,
           "frac_hbm_moved": round(moved / HBM_PEAK_GBS, 5) if moved is not None else None, "peak_GB/s": HBM_PEAK_GBS,
           "traffic_over_algorithmic": d_.get("traffic_over_algorithmic")}
    valu_bound = "valu" in d_ and d_.get("traffic_over_algorithmic") is not None and d_["traffic_over_algorithmic"] < 1.0
    head = ({"bound": "valu", "achieved": d_["valu"]["Ginst/s"], "peak": d_["valu"]["peak_Ginst/s"], "unit": "Gwave-inst/s", "frac": d_["valu"]["busy_frac"]} if valu_bound
            else {"bound": "hbm", "achieved": d_["GB/s"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": d_["frac"]})
    roofline = {**head, "kernel": f"k_{dom}", "hbm": hbm, "frac_credited": hbm["frac_credited"], "frac_hbm_moved": hbm["frac_hbm_moved"],
                "traffic": alone[dom].get("traffic"), "bytes_per_launch": alone[dom]["bytes_per_launch"], "avg_launch_ms": alone[dom]["avg_launch_ms"],
                "launches": alone[dom]["launches"], "measured_in": source, "render_lanes": lanes,
                "dominant_rule": "largest summed time among extend / shade / shadow in the exclusive pass; within 15 % the order extend > shade > shadow decides",
                "stages": {k: alone[k] for k in STAGES if k in alone}, "per_kernel": alone,
                "valu": dict(alone[dom].get("valu", {}), bound="valu", kernel=f"k_{dom}", sclk_mhz=round(sclk, 1) if sclk else None, n_simd=n_simd) if "valu" in alone[dom] else None,
                "sclk_mhz": round(sclk, 1) if sclk else None,
                "counters_source": (counters or {}).get("source"), "counters_note": counters_note,
                "pipeline_GB/s": round(sum(kb.values()) / (st["render_ms"] * 1e-3) / 1e9, 1) if st["render_ms"] > 0 else 0.0,
                "pipeline_frac": round(sum(kb.values()) / (st["render_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if st["render_ms"] > 0 else 0.0,
                "bytes_per_sample": round(sum(kb.values()) / max(1, st["n_samples"]), 1)}
    if alone is not timed:
        roofline["one_lane_Msamples/s"] = one_lane_rate
        roofline["timed_region"] = dict(prof or {}, per_kernel=timed, overlap=overlap, note="the timed region's K steps repeated with per-launch events (the timed region itself carries none)")

    out = {
        "metric": "Msamples/s (W*H*spp/s), " + ("volumetric path tracing (homogeneous media)" if volumetric else "unidirectional MIS path tracing"), "value": round(value, 3), "unit": "Msamples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32",
        "data": ("synthetic stand-in scene (adapt_amd/synth.py; the reference ships no assets for it)" if sdir == "synth" else "bundled Cornell scene file (same inputs as the reference's); no dataset involved"),
        "config": {"workload": label, "scene": f"scenes/{sdir}/{sfile}", "width": W, "height": H, "spp_per_step": spp_step,
                   "max_bounce": bounces, "num_shadow_ray": rdr.num_shadow_ray, "tiling": f"{world} x interleaved {BAND_WIDTH}-column bands",
                   "spp_per_batch": info["spp_per_batch"], "sub_queues": info["n_subqueues"], "shade_variant": info["shade_variant"], "traversal": info["traversal"],
                   "queue_MiB": round(info["queue_bytes"] / 2 ** 20, 1), "render_lanes": lanes, "seed": 0, "per_launch_events": False},
        "per_sample": {k: round(st[k] / max(1, st["n_samples"]), 4) for k in ("n_extend", "n_shade", "n_shadow", "n_shadow_traced", "n_lit", "n_draws")},
        "roofline": roofline,
    }
    if per_rank is not None:
        out["per_rank"] = per_rank

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from adapt_amd.scene_pack import make_config, pack_scene
        from oracle import binding as ob
        rc = make_config(parsed[3], width=W, height=H, max_bounce=bounces, volumetric=volumetric)
        # the product's own tree returns the brute-force hit; the reference's BVH walk loses ~1.4e-4 of the hits on meshes of small
        # triangles (tests/golden/bvhref_bunnies3.npz: its own two intersectors disagree there), so parity is judged against the
        # oracle's BRUTE-FORCE intersector and the CPU rate is taken with the reference-layout BVH (what the reference would run)
        osc = ob.OracleScene(pack_scene(*parsed), rc.cam_t, build_bvh=rc.use_bvh)
        host = host_cpu_report(ob.num_threads())
        cores = host["threads_used"]                   # min(OpenMP's count, the affinity mask, the cgroup quota): what the leg can really run on
        t = time.perf_counter(); osc.render(rc, 1, threads=cores); one = time.perf_counter() - t
        t = time.perf_counter(); osc.render(rc, 1, threads=1); one_1t = time.perf_counter() - t
        host["parallel_speedup_measured"] = {"threads": cores, "speedup_over_1_thread": round(one_1t / max(one, 1e-9), 2), "note": "1 spp of the workload on 1 thread vs on threads_used: what the host really delivers"}
        n_cpu = int(max(1, min(512, round(args.cpu_seconds / max(one, 1e-3)))))
        t = time.perf_counter(); ref, cnt, ost = osc.render(rc, n_cpu, threads=cores); cpu_dt = time.perf_counter() - t
        # AdaPT itself runs this kernel on 8 CPU threads (ti.loop_config(parallelize=8), vanilla_renderer.py:35): time that too, on a
        # quarter of the sample so the leg stays bounded
        n8 = max(1, min(n_cpu, int(round(n_cpu * min(cores, 8) / cores / 2)) or 1))
        t = time.perf_counter(); osc.render(rc, n8, threads=min(cores, 8)); dt8 = time.perf_counter() - t
        out["cpu_baseline"] = {"value": round(W * H * n_cpu / cpu_dt / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
                               "at_8_threads": {"value": round(W * H * n8 / dt8 / 1e6, 4), "spp": n8, "seconds": round(dt8, 1), "threads": min(cores, 8)},
                               "host": host,
                               "sample": f"{W}x{H} x {n_cpu} spp of the same workload (same scene, bounces, seed), {cpu_dt:.1f} s on {cores} OpenMP threads; "
                                         "oracle/pt_oracle.c = C restatement of the reference path (real AdaPT needs taichi, absent here)"
                                         + ("; reference-layout BVH" if rc.use_bvh else "")}
        vs = "cpu_baseline render (same pixels, samples, Philox stream)"
        if rc.use_bvh:                               # a window of the image, brute force (bounded: ~10 s of CPU)
            rc_b = make_config(parsed[3], width=W, height=H, max_bounce=bounces, volumetric=volumetric)
            rc_b.use_bvh = False
            wx, wy = min(W, 160), min(H, 120)
            rc_b.do_crop, rc_b.start_x, rc_b.end_x, rc_b.start_y, rc_b.end_y = True, (W - wx) // 2, (W - wx) // 2 + wx, (H - wy) // 2, (H - wy) // 2 + wy
            n_par = max(1, min(n_cpu, 8))
            ref_b, cnt_b, _ = osc.render(rc_b, n_par, threads=cores)
            win = (slice(rc_b.start_x, rc_b.end_x), slice(rc_b.start_y, rc_b.end_y))
            chk = Renderer(*parsed, width=W, height=H, max_bounce=bounces, device=local_rank)
            chk.render(n_spp=n_par)
            a, b = chk.pixels.to_numpy().astype(np.float64)[win], (ref_b / np.float32(cnt_b)).astype(np.float64)[win]
            vs = f"oracle with its BRUTE-FORCE intersector on the central {wx}x{wy} window (same pixels, samples, Philox stream)"
            n_cmp = n_par
        else:
            chk = Renderer(*parsed, width=W, height=H, max_bounce=bounces, device=local_rank)
            chk.render(n_spp=n_cpu)
            a, b = chk.pixels.to_numpy().astype(np.float64), (ref / np.float32(cnt)).astype(np.float64)
            n_cmp = n_cpu
        # upstream zeroes NaN samples but lets +-inf through (vanilla_renderer.py:119); such pixels must coincide
        fin = np.isfinite(a).all(axis=2) & np.isfinite(b).all(axis=2)
        nonfin_same = bool(np.array_equal(np.isfinite(a), np.isfinite(b)))
        af, bf = a[fin], b[fin]
        nonfin_note = None
        if not nonfin_same and not rc.use_bvh:
            # a pixel that is inf on one side only: acceptable ONLY as the zero-pdf knife-edge (oracle/binding.py explain_non_finite, DESIGN.md
            # section 5 "non-finite pixels"); anything else is a reference quirk the product build fails to reproduce, and this leg fails
            ok, findings = osc.explain_non_finite(rc, a, b, n_cmp)
            nonfin_note = {"explained_as_zero_pdf_direction_sample": ok, "pixels": findings}
            if not ok:
                print(json.dumps({"parity_failure": "non-finite pixels differ from the oracle's and are not zero-pdf knife-edges", "detail": nonfin_note}), file=sys.stderr)
                raise SystemExit(3)
        out["parity"] = {"vs": vs, "spp": n_cmp,
                         "relMSE": float(np.mean((af - bf) ** 2 / (bf ** 2 + 1e-2))), "l2_per_pixel_mean": float(np.sqrt(((af - bf) ** 2).sum(axis=1)).mean()),
                         "max_abs": float(np.abs(af - bf).max()),
                         "frac_within_1e-3": float(np.mean(np.all(np.abs(af - bf) <= 1e-3 * (1 + np.abs(bf)), axis=1))),
                         "non_finite_pixels": int((~fin).sum()), "non_finite_pixels_coincide": nonfin_same}
        if nonfin_note is not None:
            out["parity"]["non_finite_mismatch"] = nonfin_note
        out["speedup_vs_cpu"] = {"value": round(value / out["cpu_baseline"]["value"], 1), "against": f"the C port on {cores} host threads (cpu_baseline.host says what the box offers)"}
        chk.close()
    if rank == 0:
        print(json.dumps(out))
    rdr.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
