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
