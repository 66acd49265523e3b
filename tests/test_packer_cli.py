"""The corpus generator as a supported tool: `python -m pdmp3_amd.packer` writes valid streams and a
manifest; what they decode to is defined by the reference -- the compiled reference's CLI (oracle/_ref) and the
oracle's restatement agree on them byte for byte.  The reference's own main.c links against libpdmp3.so unchanged."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("what,frames", [("c1", 60), ("c3", 50), ("c4", 12), ("custom", 40)])
def test_generator_cli(oracle, tmp_path, what, frames):
    out = tmp_path / what
    cmd = [sys.executable, "-m", "pdmp3_amd.packer", what, str(out), "--frames", str(frames)]
    if what == "custom":
        cmd += ["--sfreq", "2", "--mode", "3", "--bitrate-index", "9", "--vbr", "--crc", "--seed", "77"]
    subprocess.check_call(cmd, cwd=ROOT, stderr=subprocess.DEVNULL)
    man = json.load(open(out / "manifest.json"))
    assert len(man) == (64 if what == "c4" else 1)
    for entry in man[:6] + man[-2:]:
        data = open(out / entry["file"], "rb").read()
        assert len(data) == entry["bytes"] and hashlib.sha256(data).hexdigest() == entry["sha256"]
        pcm = oracle.decode_buffer_like_cli(data)
        nch = 1 if entry["spec"]["mode"] == 3 else 2
        n = entry["spec"]["n_frames"]
        assert (n - 3) * 2304 * nch <= len(pcm) <= n * 2304 * nch       # every frame but the dropped tail (H10) decodes
    # same generator, same bytes
    subprocess.check_call(cmd[:4] + [str(tmp_path / "again")] + cmd[5:], cwd=ROOT, stderr=subprocess.DEVNULL)
    assert json.load(open(tmp_path / "again" / "manifest.json")) == man


def test_generated_stream_through_the_reference_cli(oracle, reference, tmp_path):
    """oracle/_ref/pdmp3_ref_cli = the reference's pdmp3.c + main.c compiled here: its .raw for a generated file is the
    oracle's output, byte for byte"""
    cli = os.path.join(ROOT, "oracle", "_ref", "pdmp3_ref_cli")
    if not os.path.exists(cli):
        pytest.skip("oracle/_ref/pdmp3_ref_cli not built")
    from pdmp3_amd.packer import packer
    for k, spec in enumerate([dict(n_frames=50, seed=0xC1, sfreq=0, mode=1, mode_ext=2, bitrate_index=9),
                              dict(n_frames=40, seed=5, sfreq=1, mode=0, mode_ext=0, vbr=True, block_pct=(40, 10, 40, 10))]):
        mp3 = packer.generate(**spec)
        p = tmp_path / ("s%d.mp3" % k)
        p.write_bytes(mp3)
        subprocess.check_call([cli, str(p)], timeout=120, stderr=subprocess.DEVNULL)
        assert open(str(p) + ".raw", "rb").read() == oracle.decode_buffer_like_cli(mp3)


def test_reference_main_links_against_the_library(tmp_path):
    """the drop-in claim at link level: /root/reference/main.c, unmodified, against pdmp3_amd/libpdmp3.so"""
    main_c = "/root/reference/main.c"
    if not os.path.exists(main_c):
        pytest.skip("no /root/reference here")
    from pdmp3_amd import api
    api.load_library()                                   # (builds nothing; makes sure the library is there)
    exe = str(tmp_path / "ref_main")
    libdir = os.path.join(ROOT, "pdmp3_amd")
    subprocess.check_call(["gcc", "-O2", "-w", "-o", exe, main_c, "-L" + libdir, "-lpdmp3", "-lpdmp3_hip",
                           "-Wl,-rpath," + libdir])
    syms = subprocess.check_output(["nm", "-u", exe], text=True)
    assert " U pdmp3" in syms
    r = subprocess.run([exe], capture_output=True)        # no arguments: main.c returns 1 before touching the library
    assert r.returncode == 1
