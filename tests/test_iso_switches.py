"""The ISO-correct switches (include/pdmp3.h pdmp3_amd_set_quirks, include/pdmp3_hip.h PDMP3_GC_ISO_*).

PARITY UNPINNED.  The reference has none of these modes -- it departs from ISO 11172-3 in five places
and the library reproduces them by default.  What is checked here is that every layer of the library implements the
switches THE SAME WAY as the oracle's restatement of them (oracle/pdmp3_oracle*.c: the standard's table B for
count1table_select = 1, MS up to the larger count1, intensity stereo on short blocks by the standard, scalefactor 0 for
long band 21 / short band 12), that mask 0 is still the reference bit for bit, and that each switch changes something
on the streams that exercise it.  No GPU: host stage, device stage and kernel as their host builds."""
import numpy as np
import pytest

import corpus
from pdmp3_amd.packer import packer
from test_pipeline_emul import emul_decode, emul_decode_granules, emul_decode_ring
from test_unpack_emul import emul_unpack
from util import assert_pcm_close, nch_of

ISO_TABLE33, ISO_MS_BOUND, ISO_IS_SHORT, ISO_SF21, ISO_SF12, ISO_IS_BOUND, ISO_ALL = 0x01, 0x02, 0x04, 0x08, 0x10, 0x20, 0x3f


def _streams():
    return {
        # count1table_select = 1 written with the standard's table B in 60 % of the granules (H1); MS + intensity bits on
        "joint_ms_is_t33": packer.generate(n_frames=70, seed=0x150, sfreq=0, mode=1, mode_ext=3, bitrate_index=11,
                                           block_pct=(30, 10, 50, 10), mixed_pct=40, table33_pct=60, gain=(140, 155)),
        "joint_ms_480_t33": packer.generate(n_frames=60, seed=0x151, sfreq=1, mode=1, mode_ext=2, bitrate_index=12,
                                            block_pct=(40, 15, 30, 15), mixed_pct=50, table33_pct=40, gain=(140, 155)),
        "mono_320_short": packer.generate(n_frames=50, seed=0x152, sfreq=2, mode=3, bitrate_index=8,
                                          block_pct=(20, 10, 60, 10), table33_pct=30, gain=(140, 155)),
    }


@pytest.fixture(scope="module")
def streams():
    return _streams()


@pytest.mark.parametrize("iso", [0, ISO_TABLE33, ISO_MS_BOUND, ISO_IS_SHORT, ISO_SF21, ISO_SF12, ISO_ALL])
def test_host_stage_records_equal_the_oracles(oracle, streams, iso):
    """host/frame_parse.c (header, side info, reservoir, table-driven Huffman, record builder) against the oracle's bit-serial
    restatement with the same switches: records byte-identical; mask 0 = the reference's records (test_host_stage.py pins
    those to oracle/_ref)"""
    from pdmp3_amd import api
    for name, mp3 in streams.items():
        pcm, sp_o, sd_o = oracle.decode_buffer_like_cli_iso(mp3, iso, tap_frames=100)
        sp_h, sd_h = api.parse_like_cli(mp3, 100, iso)
        n = min(sp_o.shape[0], sp_h.shape[0])
        assert n >= 45, name
        assert np.array_equal(sp_h[:n], sp_o[:n]), (name, iso)
        assert np.array_equal(sd_h[:n].view(np.uint8), sd_o[:n].view(np.uint8)), (name, iso)
        if iso == 0:
            pcm0, sp0, sd0 = oracle.decode_buffer_like_cli(mp3, tap_frames=100)
            assert pcm0 == pcm and np.array_equal(sd0.view(np.uint8), sd_o.view(np.uint8))


def test_every_switch_changes_something(oracle, streams):
    """the tests above would pass with switches that do nothing: each bit must change the records or the PCM of the
    stream that exercises it -- and nothing where it does not apply"""
    mp3 = streams["joint_ms_is_t33"]
    pcm0, sp0, sd0 = oracle.decode_buffer_like_cli_iso(mp3, 0, tap_frames=100)
    for bit, what in ((ISO_TABLE33, "spectra"), (ISO_SF21, "sf21"), (ISO_SF12, "sf12"), (ISO_MS_BOUND, "iso"), (ISO_IS_SHORT, "iso")):
        pcm, sp, sd = oracle.decode_buffer_like_cli_iso(mp3, bit, tap_frames=100)
        assert pcm != pcm0, "switch %#x does not change the PCM" % bit
        if what == "spectra":
            assert not np.array_equal(sp, sp0)
        elif what == "sf21":
            assert (sd["scalefac_l"][..., 21] == 0).all() and (sd0["scalefac_l"][..., 21] != 0).any()
        elif what == "sf12":
            assert (sd["scalefac_s"][..., 12, :] == 0).all() and (sd0["scalefac_s"][..., 12, :] != 0).any()
        else:
            assert (sd["iso"] != 0).all() and (sd0["iso"] == 0).all()
    mono = streams["mono_320_short"]
    a = oracle.decode_buffer_like_cli_iso(mono, 0)
    assert oracle.decode_buffer_like_cli_iso(mono, ISO_MS_BOUND | ISO_IS_SHORT) == a        # no joint stereo in a mono stream
    # (a mono stream's one-past-the-end slots are channel 1's scalefactors, which a mono stream never writes: zero either way)
    assert oracle.decode_buffer_like_cli_iso(mono, ISO_TABLE33) != a


@pytest.mark.parametrize("iso", [ISO_TABLE33, ISO_SF21 | ISO_SF12, ISO_ALL])
def test_device_stage_builds_the_same_records(emul, streams, iso):
    """unpack_core.h (k_unpack / k_merge as their host build) from the side info + reservoir rows the scanner hands
    over, against the host stage: identical records with the switches on, for any split into windows"""
    from pdmp3_amd import api
    for name, mp3 in streams.items():
        sp_h, sd_h = api.parse_like_cli(mp3, 100, iso)
        bits, res, _ = api.parse_bits(mp3, iso)
        assert (bits["iso"] == iso).all()
        n = min(bits.shape[0], sp_h.shape[0])
        sp_d, sd_d = emul_unpack(emul, bits, res)
        assert np.array_equal(sp_d[:n], sp_h[:n]), (name, iso)
        assert np.array_equal(sd_d[:n].view(np.uint8), sd_h[:n].view(np.uint8)), (name, iso)
        sp_c, sd_c = emul_unpack(emul, bits, res, [0, 1, n // 3, n // 2, bits.shape[0]])
        assert np.array_equal(sd_c.view(np.uint8), sd_d.view(np.uint8)) and np.array_equal(sp_c, sp_d), (name, iso)


@pytest.mark.parametrize("name", list(corpus.ISO_CASES))
def test_kernel_against_the_oracles_transforms(oracle, emul, name):
    """decode_core.h (host build of the kernels) on records that carry PDMP3_GC_ISO_* against the oracle's transforms
    with the same bits: stages 0-2 bit-exact, PCM within 1 LSB; chunk, granule and persistent kernels alike; and the
    bits matter (except on the plain-stereo case, where they must not)"""
    sp, sd = corpus.case(name)
    assert (sd["iso"] != 0).all() or name.startswith("ref_")
    want, ws = oracle.decode(sp, sd, stages=True)
    pcm, stg = emul_decode(emul, sp, sd, stages=True)
    nch = nch_of(sd)
    for k in range(3):
        assert np.array_equal(ws[:, :, :nch, k].view(np.uint32), stg[:, :, :nch, k].view(np.uint32)), "stage %d" % k
    assert_pcm_close(pcm, want, 1, name)
    whole = emul_decode(emul, sp, sd, 0)
    assert np.array_equal(whole, pcm)
    assert np.array_equal(emul_decode(emul, sp, sd, 3), whole)
    ng = 64 if "ext_varies" in name else 24
    assert np.array_equal(emul_decode_granules(emul, sp[:ng], sd[:ng]), emul_decode(emul, sp[:ng], sd[:ng], 0))
    assert np.array_equal(emul_decode_ring(emul, sp[:40], sd[:40], 9), emul_decode(emul, sp[:40], sd[:40], 0))
    if name.startswith("ref_"):
        return
    sd0 = sd.copy()
    sd0["iso"] = 0
    same = np.array_equal(oracle.decode(sp, sd0), want)
    assert same == (name == "iso_bits_on_stereo_plain"), "the record's ISO bits %s" % ("change plain stereo" if not same else "change nothing")
