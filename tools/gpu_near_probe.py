"""Where the product build's vertex-count bias comes from: on the same Philox stream the product build shades
-2.3e-4 (textured) / +1.6e-4 (features_a) vertices relative to the exact build, with the same sign under every seed.  Hypothesis: the
difference is made of rays that RE-HIT THE SURFACE THEY START ON (a grazing ray whose computed height over its own plane is rounding
noise: t = noise / cosine passes the 1e-4 threshold or not by the last bits of the intersector) and of what those paths shade afterwards.
Diagnostic builds (-DAPT_NEAR_STATS=1: tools/build_variant.sh near / EXACT=1 ... nearx) count the shaded vertices that sit within 2e-3
of the vertex before them; this script renders both builds on the same stream and prints both differences side by side.

    ADAPT_MI_LIB=build_exp/libadapt_mi_near.so ADAPT_MI_LIB_EXACT=build_exp/libadapt_mi_nearx.so python tools/gpu_near_probe.py <scene dir> <xml> <spp>"""
import os, re, subprocess, sys
if len(sys.argv) > 4:                                    # child: one build, one seed; the C side prints the counters on stderr
    sys.path.insert(0, ".")
    from adapt_amd.parsers import scene_parsing
    from adapt_amd.renderer import Renderer
    sdir, xml, spp, exact, seed = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4] == "1", int(sys.argv[5])
    r = Renderer(*scene_parsing(sdir, xml), width=64, height=48, exact=exact, seed=seed)
    r.render(n_spp=spp); r.stats(); r.close()
    sys.exit(0)
sdir, xml, spp = sys.argv[1], sys.argv[2], sys.argv[3]
res = {}
for seed in (0, 1):
    for exact in (0, 1):
        out = subprocess.run([sys.executable, __file__, sdir, xml, spp, str(exact), str(seed)], capture_output=True, text=True)
        m = re.search(r"shaded vertices (\d+), of which within 2e-3 of the vertex before them (\d+)", out.stderr)
        if not m:
            sys.exit("no [near stats] line: build the diagnostic variants first\n" + out.stderr[-600:])
        res[(exact, seed)] = (int(m.group(1)), int(m.group(2)))
for seed in (0, 1):
    (fs, fn), (es, en) = res[(0, seed)], res[(1, seed)]
    print(f"{xml} seed {seed}: shaded vertices product {fs} exact {es} diff {fs - es:+d} ({(fs - es) / es:+.2e}) | near vertices product {fn} exact {en} diff {fn - en:+d} "
          f"({(fn - en) / max(1, es):+.2e} of all vertices; near share {en / es:.2e}) | vertex diff per near diff {((fs - es) / (fn - en)) if fn != en else float('nan'):.2f}")
