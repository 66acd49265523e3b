"""Synthetic gc-record corpora for the parity tests.

Deterministic (numpy's legacy MT19937 RandomState) so that the committed golden
fixtures (tests/golden/, made by tools/make_golden.py from the compiled
reference) can be re-derived from (name, seed) alone.

Records obey the boundary contract of include/pdmp3_hip.h: spectra are zero at
and above count1 (what the reference's Read_Huffman guarantees) and the two
out-of-bounds scalefactor slots carry what the reference's memory layout would
yield, via resolve_aliases().
"""
import numpy as np

from oracle.oracle import (SIDE_DTYPE, GC_SCALEFAC_SCALE, GC_PREFLAG, GC_WIN_SWITCH, GC_MIXED,
                           GC_BLOCK_TYPE_SHIFT, FR_MODE_SHIFT, FR_MODEEXT_SHIFT, FR_RESET, SF_PEEK)

MODE_STEREO, MODE_JOINT, MODE_DUAL, MODE_MONO = 0, 1, 2, 3


def resolve_aliases(side):
    """side: array [n][2][2] of SIDE_DTYPE with scalefac_l[:21], scalefac_s[:12] filled."""
    for f in range(side.shape[0]):
        s = side[f]
        for g in range(2):
            s[g, 0]["scalefac_l"][21] = s[g, 1]["scalefac_l"][0]
            s[g, 0]["scalefac_s"][12] = s[g, 1]["scalefac_s"][0]
        s[0, 1]["scalefac_l"][21] = s[1, 0]["scalefac_l"][0]
        s[0, 1]["scalefac_s"][12] = s[1, 0]["scalefac_s"][0]
        s[1, 1]["scalefac_l"][21] = s[0, 0]["scalefac_s"][0][0]
        s[1, 1]["scalefac_s"][12] = SF_PEEK
    return side


def make_frames(n, seed, mode=MODE_JOINT, mode_ext=2, sfreq=0, block_mix=(70, 10, 10, 10),
                mixed_prob=0.5, count1_range=(0, 576), gain_range=(120, 170), big_prob=1 / 200.0,
                max_small=15, sf_max=8, reset_every=0, zero_gc_prob=0.05, is_pos_max=8, iso=0, mode_ext_choices=None):
    """Random frames.  block_mix = percentages of block types 0,1,2,3.  mode_ext_choices: the frame's mode_extension is
    drawn from these (frames with and without intensity stereo in one stream: the kernels' two copies of their code)"""
    rs = np.random.RandomState(seed)
    spectra = np.zeros((n, 2, 2, 576), dtype=np.int16)
    side = np.zeros((n, 2, 2), dtype=SIDE_DTYPE)
    cum = np.cumsum(block_mix)
    for f in range(n):
        if mode_ext_choices is not None:
            mode_ext = int(mode_ext_choices[rs.randint(0, len(mode_ext_choices))])
        fr = (sfreq & 3) | (mode << FR_MODE_SHIFT) | (mode_ext << FR_MODEEXT_SHIFT)
        if f == 0 or (reset_every and f % reset_every == 0):
            fr |= FR_RESET
        for g in range(2):
            for c in range(2):
                s = side[f, g, c]
                s["frame"] = fr
                s["iso"] = iso
                lo, hi = count1_range
                count1 = int(rs.randint(lo, hi + 1))
                count1 -= count1 % 2
                if rs.rand() < zero_gc_prob:
                    count1 = 0 if rs.rand() < 0.5 else count1
                s["count1"] = count1
                s["global_gain"] = rs.randint(gain_range[0], gain_range[1] + 1)
                fl = 0
                if rs.rand() < 0.5:
                    fl |= GC_SCALEFAC_SCALE
                if rs.rand() < 0.5:
                    fl |= GC_PREFLAG
                pct = rs.randint(0, 100)
                bt = int(np.searchsorted(cum, pct, side="right"))
                bt = min(bt, 3)
                if bt:
                    fl |= GC_WIN_SWITCH | (bt << GC_BLOCK_TYPE_SHIFT)
                    if bt == 2 and rs.rand() < mixed_prob:
                        fl |= GC_MIXED
                s["flags"] = fl
                s["subblock_gain"] = rs.randint(0, 8, size=3)
                s["scalefac_l"][:21] = rs.randint(0, sf_max, size=21)
                s["scalefac_s"][:12] = rs.randint(0, min(sf_max, is_pos_max), size=(12, 3))
                mags = rs.randint(0, max_small + 1, size=576)
                big = rs.rand(576) < big_prob
                mags[big] = rs.randint(0, 8207, size=int(big.sum()))
                sign = rs.randint(0, 2, size=576) * 2 - 1
                v = (mags * sign).astype(np.int16)
                v[count1:] = 0
                spectra[f, g, c] = v
        if mode == MODE_MONO:
            spectra[f, :, 1] = 0
    resolve_aliases(side)
    return spectra, side


# name -> kwargs; every entry is a golden fixture (tools/make_golden.py) and a parity case
CASES = {
    "ms_long_441": dict(mode=MODE_JOINT, mode_ext=2, sfreq=0, block_mix=(100, 0, 0, 0)),
    "ms_mixed_blocks_441": dict(mode=MODE_JOINT, mode_ext=2, sfreq=0, block_mix=(40, 15, 30, 15)),
    "ms_short_heavy_480": dict(mode=MODE_JOINT, mode_ext=2, sfreq=1, block_mix=(10, 10, 70, 10), count1_range=(380, 576)),
    "stereo_plain_320": dict(mode=MODE_STEREO, mode_ext=0, sfreq=2, block_mix=(50, 15, 20, 15)),
    "dual_480": dict(mode=MODE_DUAL, mode_ext=2, sfreq=1, block_mix=(60, 10, 20, 10)),
    "mono_441": dict(mode=MODE_MONO, mode_ext=0, sfreq=0, block_mix=(50, 15, 20, 15)),
    "mono_320": dict(mode=MODE_MONO, mode_ext=2, sfreq=2, block_mix=(30, 10, 50, 10)),
    "ms_count1_skew": dict(mode=MODE_JOINT, mode_ext=2, sfreq=0, count1_range=(0, 576), zero_gc_prob=0.3),
    "ms_loud_clip": dict(mode=MODE_JOINT, mode_ext=2, sfreq=0, gain_range=(170, 215), big_prob=0.02),
    "ms_sf15": dict(mode=MODE_JOINT, mode_ext=2, sfreq=0, block_mix=(40, 15, 30, 15), sf_max=16),
    "ms_resets": dict(mode=MODE_JOINT, mode_ext=2, sfreq=0, block_mix=(40, 15, 30, 15), reset_every=3),
    # intensity stereo with is_pos <= 7 (the part of H3 that is well defined: larger values read past is_ratios[6])
    "is_long_441": dict(mode=MODE_JOINT, mode_ext=1, sfreq=0, block_mix=(100, 0, 0, 0), count1_range=(100, 500), sf_max=8),
    "ms_is_long_480": dict(mode=MODE_JOINT, mode_ext=3, sfreq=1, block_mix=(100, 0, 0, 0), count1_range=(100, 500), sf_max=8),
    # ... and on short / mixed blocks (Stereo_Process_Intensity_Short, P:2190-2220): the oracle equals oracle/_ref there
    "is_short_441": dict(mode=MODE_JOINT, mode_ext=1, sfreq=0, block_mix=(30, 10, 50, 10), count1_range=(100, 500), sf_max=8),
    "ms_is_short_480": dict(mode=MODE_JOINT, mode_ext=3, sfreq=1, block_mix=(20, 10, 60, 10), count1_range=(100, 500), sf_max=8),
}


# Twins of every case at a REALISTIC level: the cases above drive the synthesis 4x .. 150 000x
# past full scale (global_gain 120..170: 6-98 % of the samples clip), these stay below it -- |PCM| peaks 5-16 k, nothing
# clipped, median 80-1600 LSB -- so that every mode / rate / block mix meets the +-1 LSB and the 1e-5 (float) bar LITERALLY
# on a signal that is neither silent nor clipped.
FS_GAIN = (128, 140)
FS_GAIN_OF = {"ms_loud_clip": (122, 134)}         # big_prob 0.02: more linbits-sized lines, same level with less gain
CASES_FS = {name + "_fs": dict(kw, gain_range=FS_GAIN_OF.get(name, FS_GAIN)) for name, kw in CASES.items()}
ALL_CASES = dict(CASES, **CASES_FS)

# The ISO-correct switches (include/pdmp3_hip.h PDMP3_GC_ISO_*: 1 = MS on every line, 2 = intensity stereo on
# short blocks multiplies by the ratios, 4 = the rest of the standard's intensity stereo) as record corpora: the kernels
# against the oracle's restatement of the same switches.  What pins that restatement is not here: FFmpeg's decode of
# conforming packer streams, tests/golden/iso_*.npz (tests/test_iso_pin.py).
ISO_CASES = {
    "iso_ms_all_441": dict(CASES["ms_mixed_blocks_441"], iso=1, gain_range=FS_GAIN),
    "iso_ms_count1_skew": dict(CASES["ms_count1_skew"], iso=1, gain_range=FS_GAIN),
    "iso_is_short_441": dict(CASES["is_short_441"], iso=2, gain_range=FS_GAIN),
    "iso_ms_is_short_480": dict(CASES["ms_is_short_480"], iso=3, gain_range=FS_GAIN),
    "iso_ms_is_long_480": dict(CASES["ms_is_long_480"], iso=3, gain_range=FS_GAIN),
    "iso_bits_on_stereo_plain": dict(CASES["stereo_plain_320"], iso=7, gain_range=FS_GAIN),      # (no joint stereo: the bits change nothing)
    # round 6, PDMP3_GC_ISO_IS_STD (4): the whole of the standard's intensity stereo -- positions from the right channel,
    # bound by its last non-zero line (per window), last band included, no M/S on intensity-coded lines.  The two
    # channels draw their block shapes independently here: the right channel's counts, in the kernel and in the oracle
    "iso_std_is_long_441": dict(CASES["is_long_441"], iso=7, gain_range=FS_GAIN),
    "iso_std_ms_is_long_480": dict(CASES["ms_is_long_480"], iso=7, gain_range=FS_GAIN),
    "iso_std_is_short_441": dict(CASES["is_short_441"], iso=7, gain_range=FS_GAIN),
    "iso_std_ms_is_short_480": dict(CASES["ms_is_short_480"], iso=7, gain_range=FS_GAIN),
    # frames with and without intensity stereo side by side: the granule kernel sends the former down its whole-frame path
    # (like mono frames), the chunk kernel runs its RARE copy for chunks that hold one (decode_core.h frame_is_rare)
    "iso_std_ext_varies_441": dict(mode=MODE_JOINT, mode_ext=2, mode_ext_choices=(2, 2, 2, 3, 1, 0, 2), sfreq=0, block_mix=(40, 15, 30, 15),
                                   count1_range=(100, 576), sf_max=8, iso=7, gain_range=FS_GAIN),
    "ref_ext_varies_480": dict(mode=MODE_JOINT, mode_ext=2, mode_ext_choices=(2, 2, 3, 1, 0, 2, 2), sfreq=1, block_mix=(40, 15, 30, 15),
                               count1_range=(100, 500), sf_max=8, iso=0, gain_range=FS_GAIN),
    "iso_std_ms_is_mixed_320": dict(mode=MODE_JOINT, mode_ext=3, sfreq=2, block_mix=(20, 10, 60, 10), mixed_prob=0.8,
                                    count1_range=(20, 400), sf_max=8, iso=7, gain_range=FS_GAIN, zero_gc_prob=0.15),
}

# int16 tolerance per case, in LSB, as literal numbers.  +-1 LSB (north_star; P:2028-2031 is the step an LSB is defined
# by) for every case but ONE: ms_loud_clip is driven to 4.5e5 x full scale (99 % of its samples clip), where one ulp of
# the binary32 sums is 0.03 = 1000 LSB, and the few samples that come back inside the int16 range differ by up to 21 LSB
# between two f32 summation orders (measured, 64 frames, host build and GPU).
PCM_TOL_LSB = {name: 1 for name in ALL_CASES}
PCM_TOL_LSB["ms_loud_clip"] = 32
# float PCM tolerance per case, absolute, as literal numbers.  1e-5 (north_star) LITERALLY for every `_fs` case.  The
# loud cases leave the range where that bar can be met by ANY binary32 evaluation in another summation order -- their
# sums reach 19 .. 141 x full scale, one ulp there is 1.9e-6 .. 1.5e-5 -- so they carry the measured difference x 2
# (64 frames; = 1.5 .. 3.5 ulp of the case's amplitude); ms_is_short_480 has one intensity-stereo band that reaches
# 4.3e9 (ulp 512), ms_loud_clip 4.5e5 (ulp 0.031).
F32_TOL_ABS = {name: 1e-5 for name in ALL_CASES}
F32_TOL_ABS.update({"ms_long_441": 4e-5, "ms_mixed_blocks_441": 6e-5, "ms_short_heavy_480": 5e-5, "stereo_plain_320": 5e-5,
                    "dual_480": 1.1e-4, "mono_441": 7e-5, "mono_320": 8e-5, "ms_count1_skew": 1.2e-5, "ms_loud_clip": 0.16,
                    "ms_sf15": 2e-5, "ms_resets": 4e-5, "is_long_441": 5e-5, "ms_is_long_480": 8e-5, "is_short_441": 4e-5,
                    "ms_is_short_480": 1536.0})
N_GOLDEN = 64                                     # frames per golden fixture (>= 64 per block-type mix)


def case(name, n=N_GOLDEN, seed=None):
    kw = ALL_CASES[name] if name in ALL_CASES else ISO_CASES[name]
    if seed is None:
        seed = (sum(ord(ch) * (i + 1) for i, ch in enumerate(name)) * 2654435761) & 0x7FFFFFFF
    return make_frames(n, seed, **kw)
