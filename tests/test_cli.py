"""Host-side CLI logic (no GPU): flags of the reference driver are accepted, --config files merge, images are encoded."""
import struct
import zlib

import numpy as np
import pytest

from adapt_amd.cli import get_options, to_display, write_image


def test_reference_flags_are_accepted(tmp_path):
    o = get_options(["--scene", "cbox", "--name", "complex.xml", "--iter_num", "128", "--arch", "cuda", "--type", "pt", "--no_gui",
                     "--save_iter", "50", "-l", "-p", "--normalize", "0.99", "--img_ext", "png", "--no_watermark", "--no_save_fig"])
    assert (o.scene, o.name, o.iter_num, o.save_iter, o.load, o.profile, o.no_gui) == ("cbox", "complex.xml", 128, 50, True, True, True)
    cfgf = tmp_path / "run.conf"
    cfgf.write_text("# comment\niter_num = 64\nscene = csphere\nno_gui = true\n")
    o = get_options(["--config", str(cfgf), "--iter_num", "8"])
    assert o.iter_num == 8 and o.scene == "csphere" and o.no_gui              # command line wins over the file
    with pytest.raises(SystemExit):
        get_options(["--type", "photon-map"])


def test_only_pt_and_vpt_are_served(capsys):
    from adapt_amd.cli import main
    assert main(["--type", "bdpt"]) == 2
    assert "only the `pt` and `vpt` renderers" in capsys.readouterr().err
    assert main(["--type", "ao"]) == 2


def test_image_orientation_and_png(tmp_path):
    img = np.zeros((4, 3, 3), np.float32)          # (w=4, h=3), [x][y], y up
    img[0, 2] = (1, 0, 0)                          # x=0, top row
    img[3, 0] = (0, 2.0, np.nan)                   # x=3, bottom row, clipped / nan->0
    px = to_display(img)
    assert px.shape == (3, 4, 3) and tuple(px[0, 0]) == (255, 0, 0) and tuple(px[2, 3]) == (0, 255, 0)
    p = tmp_path / "a.png"
    write_image(img, str(p))
    blob = p.read_bytes()
    assert blob[:8] == b"\x89PNG\r\n\x1a\n" and struct.unpack(">II", blob[16:24]) == (4, 3)
    idat = blob[blob.index(b"IDAT") + 4:blob.index(b"IEND") - 8]
    raw = zlib.decompress(idat)
    assert len(raw) == 3 * (1 + 4 * 3) and raw[1:4] == b"\xff\x00\x00"
    write_image(img, str(tmp_path / "a.bmp"))
    assert (tmp_path / "a.bmp").read_bytes()[:2] == b"BM"
    pil = pytest.importorskip("PIL.Image")                     # --img_ext jpg (parsers/opts.py:25)
    write_image(img, str(tmp_path / "a.jpg"))
    with pil.open(tmp_path / "a.jpg") as im:
        assert im.format == "JPEG" and im.size == (4, 3)
    write_image(img, str(tmp_path / "a.npy"))
    assert np.array_equal(np.load(tmp_path / "a.npy"), img, equal_nan=True)
