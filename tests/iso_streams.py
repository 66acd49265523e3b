"""The streams of the ISO pin (tests/golden/iso_*.npz): conforming MPEG-1 Layer III made by the packer with iso_strict
(include/pdmp3_packer.h: window sequence long -> start -> short ... -> stop -> long, scfsi = 0 beside short blocks, one
block shape per joint-stereo granule, one mixed flag per run of short blocks, no region index past band 22, nothing
but code words inside part2_3_length) -- the constructs on which every conforming decoder must agree.  Each fixture holds
what FFmpeg's mpegaudiodec made of the stream (tools/make_iso_golden.py, build container only); the tests decode the
same bytes with PDMP3_ISO_ALL and compare.

Levels: FFmpeg's decoder here is the FIXED-point one (int16 out, rounding to nearest; its own noise against an exact
decode is <= 1.3 LSB on these streams).  It flushes coded lines below ~2^-9 LSB to zero, which moves ITS intensity-stereo
bound (the standard's is defined on the coded integers): the intensity streams keep every coded line above that
(narrow_scales, global_gain 150..165, no linbits so that nothing clips)."""

RATES = (44100, 48000, 32000)
N_FRAMES = 40

_COMMON = dict(n_frames=N_FRAMES, bitrate_index=11, iso_strict=True)
_ALL_BLOCKS = dict(block_pct=(30, 10, 50, 10), mixed_pct=40)
_IS = dict(narrow_scales=True, big_pct=0, gain=(150, 165), is_cut_pct=80)

STREAMS = {
    # name: packer.generate kwargs.  What each one exercises: H = the SURVEY hazard whose ISO switch it needs
    "iso_stereo_441": dict(_COMMON, **_ALL_BLOCKS, seed=0x6001, sfreq=0, mode=0, mode_ext=0, table33_pct=50, gain=(120, 138)),          # H1 H4 H5
    "iso_ms_441": dict(_COMMON, **_ALL_BLOCKS, seed=0x6002, sfreq=0, mode=1, mode_ext=2, table33_pct=50, is_cut_pct=60, gain=(120, 138)),  # H2 (count1 skew)
    "iso_ms_480_vbr_crc": dict(_COMMON, **_ALL_BLOCKS, seed=0x6003, sfreq=1, mode=1, mode_ext=2, vbr=True, crc=True, table33_pct=30, gain=(120, 138)),
    "iso_mono_320": dict(_COMMON, **_ALL_BLOCKS, seed=0x6004, sfreq=2, mode=3, mode_ext=0, table33_pct=50, gain=(120, 138)),
    "iso_dual_480": dict(_COMMON, **_ALL_BLOCKS, seed=0x6005, sfreq=1, mode=2, mode_ext=0, table33_pct=50, gain=(120, 138)),
    "iso_ms_short_320k": dict(_COMMON, seed=0x6006, sfreq=0, mode=1, mode_ext=2, block_pct=(10, 10, 70, 10), mixed_pct=0, gain=(125, 140)) | dict(bitrate_index=14),
    "iso_is_long_441": dict(_COMMON, **_IS, seed=0x6011, sfreq=0, mode=1, mode_ext=1, block_pct=(100, 0, 0, 0)),                         # H3 + IS_BOUND
    "iso_is_short_441": dict(_COMMON, **_IS, seed=0x6012, sfreq=0, mode=1, mode_ext=1, block_pct=(10, 10, 70, 10), mixed_pct=0),
    "iso_ms_is_441": dict(_COMMON, **_IS, **_ALL_BLOCKS, seed=0x6013, sfreq=0, mode=1, mode_ext=3, table33_pct=40),
    "iso_ms_is_480": dict(_COMMON, **_IS, **_ALL_BLOCKS, seed=0x6014, sfreq=1, mode=1, mode_ext=3, table33_pct=40),
    "iso_ms_is_320": dict(_COMMON, **_IS, **_ALL_BLOCKS, seed=0x6015, sfreq=2, mode=1, mode_ext=3, table33_pct=40),
    "iso_ms_is_mixed_441": dict(_COMMON, **_IS, seed=0x6016, sfreq=0, mode=1, mode_ext=3, block_pct=(10, 10, 70, 10), mixed_pct=100),
}

# the literal bars (LSB of int16 full scale 32767).  FFmpeg's int16 is rounded to nearest from its fixed-point sums:
# half an LSB of rounding + its arithmetic's noise.  Measured in the build container (oracle's binary32 PCM against the
# fixtures, 12 streams x 39 frames): max 1.3, rms 0.41.
TOL_F32_LSB = 2.0          # float PCM (x 32767) against FFmpeg's int16
TOL_S16_LSB = 3.0          # int16 PCM (truncated toward zero, pdmp3.c:2028: up to one more LSB) against FFmpeg's int16
RMS_LSB = 0.6


def nch_of(kw):
    return 1 if kw["mode"] == 3 else 2


def rate_of(kw):
    return RATES[kw["sfreq"]]


# ---- MPEG-2 LSF / MPEG-2.5 (last third; include/pdmp3.h PDMP3_ISO_LSF) --------------------------------------
# {stereo, mono, M/S, M/S + intensity} x the six LSF sampling frequencies x every block type (mixed blocks everywhere but
# at 8 kHz, where three short bands are 72 lines and "the first two subbands long" has no band boundary to end at), VBR
# and CRC on two of them.  tests/golden/lsf_*.npz: FFmpeg's decode, made like the iso_* fixtures.
LSF_RATES = {(1, 0): 22050, (1, 1): 24000, (1, 2): 16000, (2, 0): 11025, (2, 1): 12000, (2, 2): 8000}
LSF_N_FRAMES = 36
_LSF_COMMON = dict(n_frames=LSF_N_FRAMES, bitrate_index=8, iso_strict=True, block_pct=(30, 10, 50, 10), table33_pct=40)
LSF_STREAMS = {}
for (_v, _sf), _rate in LSF_RATES.items():
    _k = "%dk" % (_rate // 1000)
    _mixed = 0 if (_v, _sf) == (2, 2) else 40
    _seed = 0x7000 + 16 * (_v * 3 + _sf)
    _b = dict(_LSF_COMMON, version=_v, sfreq=_sf, mixed_pct=_mixed)
    LSF_STREAMS["lsf_%s_stereo" % _k] = dict(_b, seed=_seed + 1, mode=0, mode_ext=0, gain=(120, 138))
    LSF_STREAMS["lsf_%s_mono" % _k] = dict(_b, seed=_seed + 2, mode=3, mode_ext=0, gain=(120, 138), crc=(_sf == 1))
    LSF_STREAMS["lsf_%s_ms" % _k] = dict(_b, seed=_seed + 3, mode=1, mode_ext=2, gain=(120, 138), is_cut_pct=50, vbr=(_sf == 2), vbr_lo=4, vbr_hi=12)
    LSF_STREAMS["lsf_%s_msis" % _k] = dict(_b, seed=_seed + 4, mode=1, mode_ext=3, **_IS)
del _v, _sf, _rate, _k, _mixed, _seed, _b

# 24 kHz: FFmpeg's (and mpg123's) long-block band table has band 18 start at line 330 where the standard (and LAME, libmad,
# minimp3, and this engine: pdmp3_amd/csrc/lsf_tables.h) has 332 -- lines 330 / 331 take the neighbouring band's scalefactor
# there.  At the intensity-stereo fixtures' level that is up to 5 LSB on a few granules; with the oracle's table entry set
# to 330 (orc_debug_24k_330, tests only) the same fixture is within 1.2 LSB: tests/test_lsf_pin.py shows both.
LSF_TOL_F32_LSB = {name: (6.0 if name.startswith("lsf_24k") else TOL_F32_LSB) for name in LSF_STREAMS}
LSF_TOL_S16_LSB = {name: (7.0 if name.startswith("lsf_24k") else TOL_S16_LSB) for name in LSF_STREAMS}


def lsf_rate_of(kw):
    return LSF_RATES[(kw["version"], kw["sfreq"])]
