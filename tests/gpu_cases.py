"""Case bodies shared by the two GPU parity modules: tests/test_gpu_parity.py runs them on the EXACT build, tests/test_gpu_fast.py on the
PRODUCT build (whichever library `adapt_amd._lib.use()` has selected when the body runs), each with its own tolerances.  Not collected
by pytest (no `test_` prefix)."""
import os

import numpy as np

from conftest import ROOT, golden, image_metrics, record_metric, scene_from_golden
from adapt_amd.scene_pack import make_config


def volumetric_scene_vs_reference_run_and_oracle(tag, within=0.99, rel=1e-3, draws_tol=2e-3, stat_tol=5e-4):
    """One of the reference's vpt scenes / this repo's media coverage scenes (tests/golden/vptscene_<tag>.npz): (a) against the image the
    reference's own VolumeRenderer.render produced on the same Philox stream, (b) against the oracle at more samples, where the path
    structure has to agree too: vertices shaded, light samples taken, random numbers drawn."""
    from adapt_amd.renderer import VolumeRenderer
    from adapt_amd.scene_pack import pack_scene
    from oracle import binding as ob
    tup, g = scene_from_golden(tag, "vptscene")
    w, h, spp = int(g["width"]), int(g["height"]), int(g["spp"])
    r = VolumeRenderer(*tup, width=w, height=h)
    try:
        assert r.info()["shade_variant"].startswith("volumetric")
        r.render(n_spp=spp)
        m = image_metrics(r.color.to_numpy() / spp, g["accum"] / spp)
        assert m["frac_within"] >= within and m["relMSE"] <= rel, (tag, m)
        assert abs(r.stats()["n_draws"] - int(g["draws"].sum())) <= draws_tol * int(g["draws"].sum())
        r.clear(); r.render(n_spp=16)
        rc = make_config(tup[3], width=w, height=h, volumetric=True)
        ref, _, ost = ob.OracleScene(pack_scene(*tup), rc.cam_t).render(rc, 16)
        m = image_metrics(r.color.to_numpy() / 16, ref / 16)
        assert m["frac_within"] >= within and m["relMSE"] <= rel, (tag, m)
        st = r.stats()
        for k in ("n_shade", "n_shadow", "n_draws"):
            assert abs(st[k] - ost[k]) <= stat_tol * ost[k], (tag, k, st[k], ost[k])
        assert st["n_samples"] == ost["n_samples"] == w * h * 16
        # the walk only follows light samples that can contribute, the reference follows all of them
        assert 0 < st["n_track"] <= ost["n_track"] and st["n_lit"] <= ost["n_lit"] * (1 + stat_tol)
        return r.info()
    finally:
        r.close()


def grid_volume_vs_reference_run_and_oracle(name, within=0.985, rel=2e-3, draws_tol=3e-3, stat_tol=2e-3):
    """Grid volumes on the device (scenes/test/volgrid_*.xml): delta tracking in the free-path step and ratio tracking inside the light
    sampling draw from the path's own Philox stream in the reference's order, so image, vertices shaded, light samples and draw counts
    follow the oracle (and the reference-run fixture) like every other scene."""
    from adapt_amd.parsers.xml_parser import scene_parsing
    from adapt_amd.renderer import VolumeRenderer
    from adapt_amd.scene_pack import pack_scene
    from oracle import binding as ob
    g = golden(f"vptrun_{name}.npz")
    cwd = os.getcwd(); os.chdir(ROOT)
    try:
        tup = scene_parsing(os.path.join(ROOT, "scenes", "test"), name + ".xml")
        w, h, spp = int(g["width"]), int(g["height"]), int(g["spp"])
        r = VolumeRenderer(*tup, width=w, height=h)
        fs = pack_scene(*tup)
    finally:
        os.chdir(cwd)
    try:
        assert "grid volume" in r.info()["shade_variant"]
        r.render(n_spp=spp)
        m = image_metrics(r.color.to_numpy() / spp, g["accum"] / spp)
        assert m["frac_within"] >= within and m["relMSE"] <= rel, (name, m)
        assert abs(r.stats()["n_draws"] - int(g["draws"].sum())) <= draws_tol * int(g["draws"].sum())
        r.clear(); r.render(n_spp=16)
        rc = make_config(tup[3], width=w, height=h, volumetric=True)
        ref, _, ost = ob.OracleScene(fs, rc.cam_t).render(rc, 16)
        m = image_metrics(r.color.to_numpy() / 16, ref / 16)
        assert m["frac_within"] >= within and m["relMSE"] <= rel, (name, m)
        st = r.stats()
        for k in ("n_shade", "n_shadow", "n_draws"):
            assert abs(st[k] - ost[k]) <= stat_tol * ost[k], (name, k, st[k], ost[k])
        return r.info()
    finally:
        r.close()


def c4_crop_vs_brute_force_oracle(within=0.9995, rel=1e-9, draws_tol=1e-4):      # (defaults: the exact build, measured 100 %, 4.8e-13, counts equal)
    """BASELINE configs[3] stand-in at full geometry (95 050 triangles, 800 x 800), the central 160 x 120 window x 8 spp (what bench.py's
    parity leg renders), HIP (own BVH) vs the oracle's BRUTE-FORCE intersector: the HIP tree returns the brute-force hit.  (The
    reference's own BVH, as restated in the oracle, drops ~2e-4 of the hits on this scene - its node boxes are unpadded - so it is compared
    statistically, not per pixel.)"""
    from adapt_amd.renderer import Renderer
    from adapt_amd.scene_pack import pack_scene
    from adapt_amd.synth import three_bunnies
    from oracle import binding as ob
    em, arr, objs, cfg = three_bunnies()
    spp, wx, wy = 8, 160, 120
    cfg = dict(cfg); cfg["film"] = {"width": 800, "height": 800, "crop_x": 400, "crop_y": 400, "crop_rx": wx // 2, "crop_ry": wy // 2}
    r = Renderer(em, arr, objs, cfg)
    try:
        assert r.info()["traversal"] == "bvh"
        r.render(n_spp=spp)
        rc = make_config(cfg)
        assert rc.do_crop and rc.use_bvh and (rc.end_x - rc.start_x, rc.end_y - rc.start_y) == (wx, wy)
        win = (slice(rc.start_x, rc.end_x), slice(rc.start_y, rc.end_y))
        img = r.color.to_numpy()[win]
        sc = ob.OracleScene(pack_scene(em, arr, objs, cfg), rc.cam_t, build_bvh=True)
        rc.use_bvh = False
        ref, _, ost = sc.render(rc, spp, threads=ob.num_threads())
        m = image_metrics(img / spp, ref[win] / spp)
        st = r.stats()
        record_metric(f"c4 crop {wx}x{wy}x{spp} {r.info()['arithmetic']} build", dict(m, n_shade=st["n_shade"], n_shade_oracle=ost["n_shade"], n_draws=st["n_draws"], n_draws_oracle=ost["n_draws"]))
        assert st["n_samples"] == ost["n_samples"] == spp * wx * wy
        assert m["frac_within"] >= within and m["relMSE"] <= rel, m
        for k in ("n_shade", "n_shadow", "n_draws"):
            assert abs(st[k] - ost[k]) <= draws_tol * ost[k], (k, st[k], ost[k])
        # reference-layout BVH in the oracle: same picture up to its rare lost hits
        rc.use_bvh = True
        ref_bvh, _, _ = sc.render(rc, spp, threads=ob.num_threads())
        mb = image_metrics(img / spp, ref_bvh[win] / spp)
        assert mb["frac_within"] >= 0.97, mb
        return r.info()
    finally:
        r.close()


def c5_crop_vs_brute_force_oracle(cx, cy, within=0.9995, rel=1e-9, stat_tol=2e-3):      # (defaults: the exact build, measured 100 %, 1e-15)
    """BASELINE configs[4] stand-in at full geometry (285 134 triangles, 1280 x 720, 16 bounces), a 96 x 64 window x 8 spp: HIP (own BVH)
    vs the oracle's BRUTE-FORCE intersector on the same Philox stream - per pixel, plus exact sample counts and path statistics."""
    from adapt_amd.renderer import Renderer
    from adapt_amd.scene_pack import pack_scene
    from adapt_amd.synth import bunny_field
    from oracle import binding as ob
    em, arr, objs, cfg = bunny_field()
    spp = 8
    cfg = dict(cfg); cfg["film"] = {"width": 1280, "height": 720, "crop_x": cx, "crop_y": cy, "crop_rx": 48, "crop_ry": 32}
    r = Renderer(em, arr, objs, cfg)
    try:
        assert r.info()["traversal"] == "bvh"
        r.render(n_spp=spp)
        rc = make_config(cfg)
        assert rc.do_crop and rc.use_bvh and (rc.end_x - rc.start_x, rc.end_y - rc.start_y) == (96, 64)
        win = (slice(rc.start_x, rc.end_x), slice(rc.start_y, rc.end_y))
        img = r.color.to_numpy()[win]
        sc = ob.OracleScene(pack_scene(em, arr, objs, cfg), rc.cam_t, build_bvh=False)
        rc.use_bvh = False
        ref, _, ost = sc.render(rc, spp, threads=ob.num_threads())
        m = image_metrics(img / spp, ref[win] / spp)
        record_metric(f"c5 crop ({cx}, {cy}) 96x64x{spp} {r.info()['arithmetic']} build", m)
        st = r.stats()
        assert st["n_samples"] == ost["n_samples"] == spp * 96 * 64
        assert m["frac_within"] >= within and m["relMSE"] <= rel, m
        for k in ("n_shade", "n_shadow", "n_draws"):
            assert abs(st[k] - ost[k]) <= stat_tol * ost[k], (k, st[k], ost[k])
        return r.info()
    finally:
        r.close()
