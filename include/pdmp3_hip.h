/*
 * pdmp3_hip.h -- C-ABI of the MI355X (gfx950) transform engine.
 *
 * This is the drop-in boundary for the ONE data-parallel hot path of
 * technosaurus/PDMP3: everything `Decode_L3` (pdmp3.c:1024-1060) does to one
 * parsed frame, plus `Convert_Frame_S16` (pdmp3.c:2307-2345):
 *
 *   L3_Requantize (P:1829) -> L3_Reorder (P:1786) -> L3_Stereo (P:1911) ->
 *   L3_Antialias (P:1706) -> L3_Hybrid_Synthesis/IMDCT_Win (P:1752/P:1649) ->
 *   L3_Frequency_Inversion (P:1738) -> L3_Subband_Synthesis (P:1978) ->
 *   interleaved int16 PCM.
 *
 * Plain C: pointers, sizes, PODs.  No C++/torch types.  Device pointers are
 * ordinary `void*` HIP device addresses; `stream` is a `hipStream_t` passed as
 * `void*` (NULL = the default stream).  All entry points return 0 on success
 * or a negative PDMP3_HIP_E* code; pdmp3_hip_last_error() gives the text.
 *
 * "P:n" = /root/reference/pdmp3.c line n.
 */
#ifndef PDMP3_HIP_H
#define PDMP3_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PDMP3_HIP_OK        0
#define PDMP3_HIP_EINVAL   -1   /* bad argument */
#define PDMP3_HIP_EDEVICE  -2   /* HIP runtime error (see pdmp3_hip_last_error) */
#define PDMP3_HIP_ENOMEM   -3

/* ------------------------------------------------------------------------
 * The record the device needs per granule-channel ("gc").
 *
 * It replaces the reference's in-handle structs t_mpeg1_side_info (P:71-95),
 * t_mpeg1_main_data (P:96-101) and the three header fields the hot path reads
 * (mode, mode_extension, sampling_frequency; P:56-70).
 *
 * Spectra travel separately as int16 (the reference keeps Huffman integers in
 * `float is[2][2][576]`, P:99; |value| <= 8206 = 15 + 2^13 - 1, P:1637).
 * Contract (what the reference's own Read_Huffman guarantees, P:2107-2111):
 * spectra[n] == 0 for n >= count1.
 * ---------------------------------------------------------------------- */

/* flags */
#define PDMP3_GC_SCALEFAC_SCALE  0x01u  /* P:92 */
#define PDMP3_GC_PREFLAG         0x02u  /* P:91 */
#define PDMP3_GC_WIN_SWITCH      0x04u  /* P:80 */
#define PDMP3_GC_BLOCK_TYPE_SHIFT 3     /* 2 bits, P:82 (0 when win_switch==0, P:1191) */
#define PDMP3_GC_BLOCK_TYPE_MASK 0x18u
#define PDMP3_GC_MIXED           0x20u  /* P:83 */

/* frame byte (identical in all gc records of one frame) */
#define PDMP3_FR_SFREQ_MASK      0x03u  /* sampling_frequency index: 0=44100 1=48000 2=32000 (P:529) */
#define PDMP3_FR_MODE_SHIFT      2      /* 2 bits: 0 stereo 1 joint 2 dual 3 mono (P:49-55) */
#define PDMP3_FR_MODE_MASK       0x0Cu
#define PDMP3_FR_MODEEXT_SHIFT   4      /* 2 bits: bit1 = MS, bit0 = intensity (P:1918,1932) */
#define PDMP3_FR_MODEEXT_MASK    0x30u
#define PDMP3_FR_RESET           0x40u  /* zero overlap + polyphase FIFO before this frame
                                           (what hsynth_init/synth_init do after
                                           pdmp3_open_feed, P:1757-1766, P:1996-2003) */

/* scalefac_s[12][w] marker: "the reference reads the float bits of
 * is[0][0][w] here" (granule 1 / channel 1 aliases the previous
 * granule's synthesis output).  Resolved on the device. */
#define PDMP3_SF_PEEK            0xFFu

typedef struct pdmp3_gc_side {
  uint16_t count1;            /* P:93; 0..576; first line of the rzero region      */
  uint8_t  global_gain;       /* P:77                                              */
  uint8_t  flags;             /* PDMP3_GC_*                                        */
  uint8_t  subblock_gain[3];  /* P:85                                              */
  uint8_t  frame;             /* PDMP3_FR_*                                        */
  uint8_t  scalefac_l[22];    /* P:97; [21] = value the reference's out-of-bounds
                                 read yields, resolved by the host     */
  uint8_t  scalefac_s[13][3]; /* P:98; [12][w] = out-of-bounds value
                                 or PDMP3_SF_PEEK                                  */
  uint8_t  iso;               /* PDMP3_GC_ISO_*: 0 = the reference's behaviour (below)  */
  uint8_t  lsf;               /* PDMP3_LSF_*: 0 = an MPEG-1 frame (all the reference knows) */
  uint8_t  lsf_slen[4];       /* channel 1 of an LSF intensity-stereo frame: the widths ... */
  uint8_t  lsf_nsfb[4];       /* ... and the sizes of its four scalefactor partitions       */
  uint8_t  reserved[49];      /* must be zero                                      */
} pdmp3_gc_side;              /* 128 bytes                                         */

/* pdmp3_gc_side.lsf -- MPEG-2 LSF and "MPEG-2.5" frames (ISO/IEC 13818-3; SURVEY 8f #4, last third).  NOT the reference,
 * which rejects them (P:1293): there is no reference behaviour to reproduce, so an LSF frame is decoded by the standard
 * -- as if every PDMP3_GC_ISO_* bit were set, with scalefac_l[21] = scalefac_s[12][w] = 0 -- and pinned by FFmpeg's decode
 * of the same streams (tests/golden/lsf_*.npz).  What differs from an MPEG-1 record:
 *   - the frame has ONE granule: the records [1][ch] of its frame slot are ignored, the synthesis state passes over them
 *     untouched, the frame's PCM is the first HALF of its place (576 sample-frames: 2304 bytes stereo, 1152 mono);
 *   - the sampling frequency is kLsfSampleRates[3 * version + (frame & 3)] (pdmp3_amd/csrc/lsf_tables.h): six more
 *     scalefactor-band tables; a mixed block's long part is bands 0..5 (36 lines), not 0..7;
 *   - PDMP3_GC_PREFLAG is what scalefac_compress >= 500 implies (no preflag bit in the stream);
 *   - intensity stereo (13818-3 2.4.3.2): is_pos p scales ONE channel by i0^((p + 1) / 2) (p odd: left, p even: right),
 *     i0 = 2^(-1/4) or 2^(-1/2) by PDMP3_LSF_IS_SCALE; "not intensity coded" is the largest value the partition's
 *     slen can hold, so channel 1's record carries its partitions (lsf_slen / lsf_nsfb, in transmission order: long
 *     bands, or band by band x window).                                                                              */
#define PDMP3_LSF_VERSION_MASK  0x03u  /* 1 = MPEG-2 LSF (22.05 / 24 / 16 kHz), 2 = MPEG-2.5 (11.025 / 12 / 8 kHz); identical
                                          in all gc records of one frame */
#define PDMP3_LSF_IS_SCALE      0x04u  /* channel 1: intensity_scale = scalefac_compress & 1 */

/* pdmp3_gc_side.iso (identical in all gc records of one frame) -- SURVEY 8f #4, "ISO-correct switches".  The reference
 * departs from ISO 11172-3 in five places; by default this engine reproduces them bit for bit.  A caller
 * that wants the standard's behaviour sets these bits (include/pdmp3.h: pdmp3_amd_set_quirks does it for a handle).  Three
 * of the six live in the transforms and travel in the record; the other three (H1 the count1 table, H4 / H5 the
 * one-past-the-end scalefactors) are decided where the records are built: a record with scalefac_l[21] = 0 and
 * scalefac_s[12][w] = 0 IS the standard's.  Nothing in the reference pins these modes; since round 6 an independent ISO
 * decoder does (FFmpeg, through tests/golden/iso_*.npz: DESIGN.md section 4). */
#define PDMP3_GC_ISO_MS_ALL    0x01u  /* H2: MS stereo on every line (P:1920 stops at the smaller count1, counted in reordered
                                         lines; until round 5 this bit meant "below the larger count1", which in short blocks
                                         leaves coded lines of the last band unrotated) */
#define PDMP3_GC_ISO_IS_SHORT  0x02u  /* H3: intensity stereo on short blocks multiplies by the ratios of the line's own
                                         window (P:2191 holds them in `unsigned`, P:2212-2213 assign instead of multiply,
                                         P:2203 takes the window from the un-reordered position) */
#define PDMP3_GC_ISO_IS_STD    0x04u  /* round 6, found with an independent ISO decoder (FFmpeg): the whole
                                         of the standard's intensity stereo -- is_pos from the RIGHT channel's scalefactors
                                         (P:2163 / P:2200 read the left one's), the intensity region bounded by the right
                                         channel's last non-zero line, per window in short blocks (P:1946-1965 use its count1),
                                         the last band (long 21, short 12) with the position of the band below (P:1961 / P:1953
                                         leave it out), the block shape from the right channel, no M/S rotation of intensity-coded
                                         lines.  Implies the arithmetic of PDMP3_GC_ISO_IS_SHORT */

#define PDMP3_GC_LINES          576
#define PDMP3_FRAME_GCS         4                 /* [gr][ch] = 2 x 2                */
#define PDMP3_FRAME_SPECTRA_BYTES (4 * 576 * 2)   /* int16 [2][2][576]               */
#define PDMP3_FRAME_SIDE_BYTES  (4 * 128)
#define PDMP3_FRAME_PCM_BYTES   (1152 * 2 * 2)    /* stereo; mono uses the first half */

typedef struct pdmp3_hip_ctx pdmp3_hip_ctx;

/* Create an engine on HIP device `device` (uploads the constant tables:
 * P:572-870 literals + the libm-derived pow/cos tables of P:979, P:1992,
 * P:2127-2128, P:2144-2146, generated on the host with the same expressions). */
int pdmp3_hip_create(int device, pdmp3_hip_ctx** out);
void pdmp3_hip_destroy(pdmp3_hip_ctx* ctx);
/* Environment read by pdmp3_hip_create() (none of them changes a result):
 *   PDMP3_HIP_CHAIN=0             every launch takes the chunk kernel (a chunk of frames per wave, halo)
 *   PDMP3_HIP_GRAN_MAX=n          largest launch, in frames, that takes the granule kernel (default 12288 on an MI355X)
 *   PDMP3_HIP_RING_MIN=n          launches of at least n frames take the persistent granule kernel k_decode_p (default 0 = never;
 *                                 only in a library built with -DPDMP3_WITH_RING_KERNEL: the default build does not carry the kernel)
 *   PDMP3_HIP_DIRECT_MAX=n        largest batch of a stream object that runs on the pinned host buffers directly (default 32; 0: never)
 *   PDMP3_HIP_SF_HINT=0|1|2       sampling frequency whose line table the granule kernel keeps in LDS (default 0 = 44.1 kHz;
 *                                 granules of another one read the table from memory)
 *   PDMP3_HIP_DEBUG_FAR_TIMEOUT=1 tests: every wait for another workgroup gives up at once (the independent way is taken)
 *   PDMP3_HIP_UNPACK_PROF=1       development: shader-clock stamps of k_unpack's steps, one line per launch on stderr
 *   PDMP3_HIP_GRAN_W=8            development: every launch of the granule kernel in workgroups of 8 waves (84 KB of LDS: one fits a
 *                                 CU beside a workgroup of k_unpack; by default launches that fill the chip take 16 -- measured with the
 *                                 whole-stream decoder: 30.7 against 32.8 M frames/s)
 * and by the host library (libpdmp3.so): PDMP3_DEVICE, PDMP3_STREAM_THREADS, PDMP3_NO_READAHEAD (include/pdmp3.h),
 * PDMP3_BULK_HOST_HUFFMAN, PDMP3_BULK_SNAPSHOT_ROWS, PDMP3_BULK_TRACE (include/pdmp3_bulk.h), PDMP3_CLI_STREAMING, PDMP3_CLI_WAV. */
const char* pdmp3_hip_last_error(void);

/* Bytes of one stream's carried synthesis state (replaces the function-static
 * `store[2][32][18]` P:1755 and `v_vec[2][1024]` P:1983; opaque layout). */
size_t pdmp3_hip_state_bytes(void);

/*
 * Decode `n_frames` consecutive frames of ONE stream: replaces n calls of
 * Decode_L3 (P:1024) + Convert_Frame_S16 (P:2307).
 *
 *   d_spectra  device, int16 [n_frames][2 gr][2 ch][576]
 *   d_side     device, pdmp3_gc_side [n_frames][2][2]
 *   d_pcm      device, int16; frame f occupies bytes [f*4608, f*4608+4608)
 *              (mono frames: the first 2304 bytes are used)
 *   d_state    device, pdmp3_hip_state_bytes() bytes, read before the first
 *              frame and written after the last one; NULL = start from the
 *              zero state and discard the final state
 *   chunk_frames  frames per wavefront ("chunk"); 0 = choose automatically.
 *              Chunks of several frames are independent: each re-derives the
 *              state at its start from a halo of preceding frames.
 *              PDMP3_HIP_CHUNK_PERSISTENT (-3): the persistent granule kernel (k_decode_p; tests and tools: it is
 *              bit-identical and, as measured in round 4, slower than the engine's own choice at every size).
 *              At 0 or 1, launches of up to 12288 frames (MI355X; PDMP3_HIP_GRAN_MAX)
 *              are decoded ONE GRANULE PER WAVEFRONT instead: the wavefronts hand
 *              IMDCT tails and polyphase rows on, no halo (scratch is kept per HIP
 *              stream -- for up to 32 streams, beyond that the halo form is used --
 *              or per pdmp3_hip_stream object).  A wavefront never depends on
 *              another workgroup for progress: that wait is bounded and ends in a
 *              halo decode.  PDMP3_HIP_CHAIN=0 in the environment at
 *              pdmp3_hip_create() keeps the halo form everywhere.  Same PCM and
 *              state either way, bit for bit.  pdmp3_hip_last_launch_kind() tells
 *              which form the engine's latest launch took.
 *
 * Asynchronous on `stream`.
 */
#define PDMP3_HIP_CHUNK_PERSISTENT (-3)   /* chunk_frames: take the persistent granule kernel whatever the launch size (tests, tools);
                                             PDMP3_HIP_EINVAL from a library built without -DPDMP3_WITH_RING_KERNEL (the default) */
int pdmp3_hip_decode_frames(pdmp3_hip_ctx* ctx,
                            const int16_t* d_spectra,
                            const pdmp3_gc_side* d_side,
                            int n_frames,
                            void* d_state,
                            int16_t* d_pcm,
                            int chunk_frames,
                            void* stream);

/* The granule kernel's hand-over scratch for bare calls is kept per HIP stream: 17 KB per frame of the largest such
 * launch seen on the stream (207 MB at the 12288-frame ceiling), for up to 32 streams, least recently used first out.
 * A caller that is done with a HIP stream -- or creates many short-lived ones -- gives the scratch back with this call
 * (it waits for the stream's launches).  pdmp3_hip_stream objects own theirs and free it when destroyed. */
int pdmp3_hip_release_stream_scratch(pdmp3_hip_ctx* ctx, void* stream);

/* How the latest decode launch of this engine (any thread) was laid out:
 * PDMP3_HIP_LAUNCH_CHUNKS = independent chunks with halos (k_decode), otherwise
 * the granule kernel (k_decode_g) with 8 / 16 wavefronts per workgroup.
 * For reports (bench.py names the kernel it timed from this), not for control. */
#define PDMP3_HIP_LAUNCH_NONE      0
#define PDMP3_HIP_LAUNCH_CHUNKS    1
#define PDMP3_HIP_LAUNCH_GRANULES8  8
#define PDMP3_HIP_LAUNCH_GRANULES16 16
#define PDMP3_HIP_LAUNCH_PERSISTENT 32   /* k_decode_p: workgroups of 16 wavefronts going round a range of frames each */
int pdmp3_hip_last_launch_kind(const pdmp3_hip_ctx* ctx);
/* The device's PCI address ("0000:c1:00.0") into buf (len >= 16): what a host stage needs to find the CPUs and the
 * memory next to the GPU (/sys/bus/pci/devices/<address>/local_cpulist, numa_node) -- the whole-stream decoder keeps
 * its helper threads and its pinned buffers there (include/pdmp3_bulk.h).  No reference counterpart. */
int pdmp3_hip_pci_bus_id(const pdmp3_hip_ctx* ctx, char* buf, int len);

/* Float PCM (not in the reference, whose only output is int16): the same decode, but what is stored is
 * the binary32 synthesis sum that P:2028-2031 scale by 32767, truncate and clip -- full scale is +-1.0, nothing is
 * clipped.  d_pcm: device, float; frame f occupies floats [f*2304, f*2304+2304), interleaved L, R (mono frames: the
 * first 1152).  The int16 output of pdmp3_hip_decode_frames is exactly clip(trunc(these * 32767)), which is how the
 * float form is pinned to the reference.  State and chunking as above (the two forms may alternate on one state). */
#define PDMP3_FRAME_PCM_F32_BYTES (1152 * 2 * 4)
int pdmp3_hip_decode_frames_f32(pdmp3_hip_ctx* ctx,
                                const int16_t* d_spectra,
                                const pdmp3_gc_side* d_side,
                                int n_frames,
                                void* d_state,
                                float* d_pcm,
                                int chunk_frames,
                                void* stream);

/* MPEG-2 LSF / MPEG-2.5 frames (pdmp3_gc_side.lsf != 0, above; SURVEY 8f #4 -- nothing in the reference: it rejects the
 * streams, pdmp3.c:1293).  n_frames consecutive frames of ONE stream, all LSF and all of one channel count; records and
 * spectra in the usual layout ([n_frames][2][2], of which only [0][ch] is read: an LSF frame is one granule).  The PCM
 * comes out in stream order, 576 sample-frames per frame: stereo frame f at bytes [f * 2304, + 2304); mono frame f at
 * byte (f / 2) * 4608 + (f % 2) * 1152, 1152 bytes (a PAIR of mono frames in the first half of a 4608-byte place, like
 * the granules of an MPEG-1 mono frame).  _f32: the same places counted in floats.  d_state as for
 * pdmp3_hip_decode_frames, and the same state block: MPEG-1 and LSF launches may follow each other on it. */
int pdmp3_hip_decode_lsf_frames(pdmp3_hip_ctx* ctx, const int16_t* d_spectra, const pdmp3_gc_side* d_side,
                                int n_frames, void* d_state, int16_t* d_pcm, void* stream);
int pdmp3_hip_decode_lsf_frames_f32(pdmp3_hip_ctx* ctx, const int16_t* d_spectra, const pdmp3_gc_side* d_side,
                                    int n_frames, void* d_state, float* d_pcm, void* stream);

/* Same, additionally dumping float32 stage outputs for parity tests
 * (d_stages: float [n_frames][2][2][4][576]; stage 0 = after requantize +
 * reorder, 1 = after stereo, 2 = after antialias, 3 = after hybrid synthesis
 * + frequency inversion).  Single-workgroup, slow; test use only. */
int pdmp3_hip_decode_frames_stages(pdmp3_hip_ctx* ctx,
                                   const int16_t* d_spectra,
                                   const pdmp3_gc_side* d_side,
                                   int n_frames,
                                   void* d_state,
                                   int16_t* d_pcm,
                                   float* d_stages,
                                   void* stream);

/*
 * Synthetic-workload generator of SURVEY 8d (C2 / C5), counter-based
 * splitmix64 so every GPU can generate its own shard on device:
 * frames [first_frame, first_frame + n_frames) of the stream `seed`.
 * The same integer-only generator exists on the host in the oracle
 * (oracle/pdmp3_oracle.c: orc_generate_frames); tests check they agree.
 */
int pdmp3_hip_generate_frames(pdmp3_hip_ctx* ctx,
                              uint64_t seed,
                              int64_t first_frame,
                              int n_frames,
                              int16_t* d_spectra,
                              pdmp3_gc_side* d_side,
                              void* stream);

/* ------------------------------------------------------------------------
 * Host-buffer streaming helper: what the libmpg123-style API (include/pdmp3.h,
 * pdmp3_read) drives.  One pdmp3_hip_stream = one decoder handle's device
 * state (overlap + polyphase history), a HIP stream and pinned, library-owned
 * staging buffers; the sequential Huffman stage on the host fills gc records,
 * this call moves them with hipMemcpyAsync, runs the transforms and brings the
 * PCM back.  Replaces `Decode_L3(id)` at pdmp3.c:2453 for a batch of frames.
 * ---------------------------------------------------------------------- */
typedef struct pdmp3_hip_stream pdmp3_hip_stream;

int pdmp3_hip_stream_create(pdmp3_hip_ctx* ctx, int max_frames, pdmp3_hip_stream** out);
void pdmp3_hip_stream_destroy(pdmp3_hip_stream* hs);
/* zero the carried synthesis state (pdmp3_open_feed, pdmp3.c:2377-2378) */
int pdmp3_hip_stream_reset(pdmp3_hip_stream* hs);
/* pinned staging buffers to fill / read: capacity max_frames frames each */
int16_t* pdmp3_hip_stream_spectra(pdmp3_hip_stream* hs);
pdmp3_gc_side* pdmp3_hip_stream_side(pdmp3_hip_stream* hs);
const int16_t* pdmp3_hip_stream_pcm(pdmp3_hip_stream* hs);
/* decode frames [0, n_frames) of the staging buffers; synchronous */
int pdmp3_hip_stream_decode(pdmp3_hip_stream* hs, int n_frames);

/* Pipelined form for bulk decoding of one long stream (the C3 / C4 corpora of
 * SURVEY 8d): `n_slots` (1..8) independent staging slots of `max_frames` frames.
 * While the host fills slot w+1, slot w is uploading / running / downloading.
 * Batches are decoded in SUBMIT order, each from the synthesis state the
 * previous one left (the same carry Decode_L3's static buffers give the
 * reference, pdmp3.c:1777, 2126), whatever slot they sit in.  The accessors
 * above and pdmp3_hip_stream_decode are slot 0. */
int pdmp3_hip_stream_create_slots(pdmp3_hip_ctx* ctx, int max_frames, int n_slots, pdmp3_hip_stream** out);
int pdmp3_hip_stream_slots(const pdmp3_hip_stream* hs);
int pdmp3_hip_stream_capacity(const pdmp3_hip_stream* hs);
int16_t* pdmp3_hip_stream_slot_spectra(pdmp3_hip_stream* hs, int slot);
pdmp3_gc_side* pdmp3_hip_stream_slot_side(pdmp3_hip_stream* hs, int slot);
const int16_t* pdmp3_hip_stream_slot_pcm(pdmp3_hip_stream* hs, int slot);
/* enqueue H2D + transforms + D2H of the slot's first n_frames frames; returns at once */
int pdmp3_hip_stream_submit(pdmp3_hip_stream* hs, int slot, int n_frames);
/* PCM of pdmp3_hip_stream_submit / _decode as float from now on (on != 0; see pdmp3_hip_decode_frames_f32) or as
 * int16 again.  Call with nothing in flight: the slots' PCM buffers are re-allocated (9216 bytes per frame for float)
 * and the pdmp3_hip_stream_*pcm accessors return the new ones, to be read as float.  Not for the _to / _bits forms. */
int pdmp3_hip_stream_set_f32(pdmp3_hip_stream* hs, int on);
/* The records of the following submits are LSF frames (on != 0: decoded like pdmp3_hip_decode_lsf_frames, PCM in its layout;
 * the _to forms take row_bytes 2304 for stereo and 1152 for mono frames then) or MPEG-1 frames again (0).  Batches are
 * homogeneous: the host splits them where the stream changes version or channel count.  Not for the _bits forms. */
int pdmp3_hip_stream_set_lsf(pdmp3_hip_stream* hs, int on);
/* undo the slot's latest pdmp3_hip_stream_submit beyond its first keep_frames frames: the carried synthesis state
 * (P:1755, P:1983) becomes what it was after frame keep_frames - 1 of that batch.  Blocks.  (pdmp3_read hands frames
 * out in the reference's order; frames it decoded ahead that the reference turns out not to reach are taken back.) */
int pdmp3_hip_stream_rewind(pdmp3_hip_stream* hs, int slot, int keep_frames);

/* ------------------------------------------------------------------------
 * Bitstream-level input: scalefactor + Huffman decoding on the
 * device.  The host keeps only the strictly sequential part of Read_Frame
 * (pdmp3.c:1217-1244: header sync, side info P:1129-1200, bit reservoir
 * P:1096-1122) and hands over, per frame, the side info and a snapshot of the
 * reservoir buffer g_main_data_vec (P:137) as Get_Main_Data left it.  One lane
 * per granule-channel then does what Read_Main_L3 (P:1376-1437), Read_Huffman
 * (P:2051-2115) and Huffman_Decode (P:1593-1643) do, a second small kernel
 * carries scalefactors / count1 from frame to frame exactly as the reference's
 * never-cleared g_main_data / g_side_info do, and the transform
 * kernel runs on the records in place: decoded spectra never cross PCIe
 * (2144 B per frame up instead of 5120).
 * ---------------------------------------------------------------------- */
#define PDMP3_RESERVOIR_BYTES 2064          /* 2048 + 16: one row per frame              */

typedef struct pdmp3_gc_bits {              /* side info of one granule-channel, P:74-93 */
  uint16_t part2_3_length;                  /* P:75 */
  uint16_t big_values;                      /* P:76 */
  uint8_t  global_gain;                     /* P:77 */
  uint8_t  scalefac_compress;               /* P:78 */
  uint8_t  flags;                           /* PDMP3_GC_* exactly as in pdmp3_gc_side    */
  uint8_t  table_select[3];                 /* P:84 */
  uint8_t  subblock_gain[3];                /* P:85 */
  uint8_t  region0_count, region1_count;    /* P:86-87 (the implicit 8 / 7 and 20 - r0 of
                                               P:1177-1180 already filled in)            */
  uint8_t  count1table_select;              /* P:90 */
} pdmp3_gc_bits;                            /* 16 bytes */

/* pdmp3_frame_bits.frame only: the parse state that survives frames (scalefactors, count1: SURVEY H4-H6) is zero
 * before this frame -- what a freshly allocated handle of the reference starts with.  Lets several independent
 * streams follow each other through one pdmp3_hip_stream without a host-side reset in between (the synthesis
 * state has PDMP3_FR_RESET for that).  Not copied into the gc records. */
#define PDMP3_FR_NEWSTREAM       0x80u

typedef struct pdmp3_frame_bits {
  uint8_t  frame;                           /* PDMP3_FR_* (+ PDMP3_FR_NEWSTREAM) */
  uint8_t  scfsi[2];                        /* [ch]: bit b = band group b reuses granule 0, P:73 */
  uint8_t  iso;                             /* PDMP3_ISO_* of include/pdmp3.h for this frame (0 = the reference's behaviour):
                                               TABLE33 selects the code book, SF21 / SF12 keep the one-past-the-end
                                               scalefactor slots of the records zero, MS_BOUND / IS_SHORT / IS_BOUND
                                               become the records' PDMP3_GC_ISO_* bits */
  uint8_t  reserved[12];
  pdmp3_gc_bits gc[4];                      /* [gr][ch] */
} pdmp3_frame_bits;                         /* 80 bytes */

/* pinned staging of a slot: n frames of side info and n reservoir rows */
pdmp3_frame_bits* pdmp3_hip_stream_slot_bits(pdmp3_hip_stream* hs, int slot);
uint8_t* pdmp3_hip_stream_slot_reservoir(pdmp3_hip_stream* hs, int slot);
/* like pdmp3_hip_stream_submit, from bits: H2D, unpack, merge, transforms, D2H of the PCM */
int pdmp3_hip_stream_submit_bits(pdmp3_hip_stream* hs, int slot, int n_frames);
/* Pinned host memory for PCM that should not be copied twice: with a destination inside such an allocation -- or in
 * device memory, for consumers on the GPU -- the _to forms below move a batch's PCM straight to it (row_bytes 4608, or 2304 to pack mono frames densely)
 * instead of into the slot's staging buffer; pdmp3_hip_stream_wait() then means "it is there". */
int pdmp3_hip_host_alloc(size_t bytes, void** out);
void pdmp3_hip_host_free(void* p);
int pdmp3_hip_host_is_pinned(const void* p, size_t bytes);          /* [p, p + bytes): 1 = pinned host memory, 2 = device memory
                                                                       (both are valid _to destinations), 0 = neither */
/* plain copy of `bytes` from host memory to a destination pdmp3_hip_host_is_pinned() classified (1 or 2); blocks until
 * it is done.  The whole-stream decoder uses it for the windows that cannot go to a DEVICE destination directly
 * (mixed mono / stereo frames, the clipped tail): they are staged in the slot's pinned buffer and copied from there. */
int pdmp3_hip_copy_to_dest(void* dst, const void* src_host, size_t bytes);
int pdmp3_hip_stream_submit_to(pdmp3_hip_stream* hs, int slot, int n_frames, void* pinned_dst, int row_bytes);
int pdmp3_hip_stream_submit_bits_to(pdmp3_hip_stream* hs, int slot, int n_frames, void* pinned_dst, int row_bytes);

/* Compact form of the same input.  A frame's reservoir buffer is the last main_data_begin bytes of what was there
 * before plus the frame's own main data (Get_Main_Data, P:1096-1122), so consecutive snapshots overlap almost
 * entirely: instead of 2064 bytes per frame the host uploads a POOL -- the main-data bytes of the window's frames,
 * each appended once, in stream order -- and a descriptor per frame, and the device rebuilds the rows
 * (k_rows; unpack_core.h row_byte) before k_unpack reads them.  Byte j of frame f's buffer is
 *     pool[row_off(f) + j]                 j <  top(f)       written by this frame's Get_Main_Data
 *     pool[row_off(g) + j]                 j >= top(f)       left there by frame g, the nearest earlier frame of the
 *                                                            same SEGMENT with top(g) > j (the reference never clears
 *                                                            g_main_data_vec: corrupt streams read those bytes); g is
 *                                                            found along the `up` links, each to a larger top
 *     pool[s_off + j]                      no such frame     the buffer as it was before the segment began
 * A segment is a run of frames over which "the buffer's valid bytes are the tail of the pool" holds; it starts with a
 * 2064-byte image of the buffer (at s_off) followed by a copy of its last bytes, and a new one starts at every
 * window and after anything irregular (reservoir underflow H9, short reads H18).  A frame whose buffer does not fit
 * the rule carries its own image: row_off = s_off, top = 2064.  ~1.1 KB per frame up instead of 2144. */
typedef struct pdmp3_row_desc {
  uint32_t row_off;                         /* pool offset of byte 0 of the frame's buffer            */
  uint32_t s_off;                           /* pool offset of its segment's 2064-byte image            */
  uint16_t top;                             /* bytes [0, top) come from row_off                        */
  uint16_t back;                            /* frames before this one in its segment                   */
  uint16_t up;                              /* distance (in frames) to the nearest earlier frame of the
                                             * segment with a larger top; 0 = none (the image at s_off) */
  uint16_t reserved;
} pdmp3_row_desc;                           /* 16 bytes */
pdmp3_row_desc* pdmp3_hip_stream_slot_rowdesc(pdmp3_hip_stream* hs, int slot);
uint8_t* pdmp3_hip_stream_slot_pool(pdmp3_hip_stream* hs, int slot);       /* = pdmp3_hip_stream_slot_reservoir's memory */
#define PDMP3_POOL_SLACK_BYTES 8192         /* a window of ONE frame still takes a segment start + its frame + an image */
size_t pdmp3_hip_stream_pool_bytes(const pdmp3_hip_stream* hs);            /* capacity of a slot's pool:
                                                                            * max_frames * 2064 + PDMP3_POOL_SLACK_BYTES */
/* like pdmp3_hip_stream_submit_bits_to, from the slot's bits, descriptors and the first pool_bytes of its pool */
int pdmp3_hip_stream_submit_pool_to(pdmp3_hip_stream* hs, int slot, int n_frames, size_t pool_bytes, void* pinned_dst, int row_bytes);

/* test hook: the gc records the device built for the slot's last submit_bits (after pdmp3_hip_stream_wait) */
int pdmp3_hip_stream_fetch_records(pdmp3_hip_stream* hs, int slot, int n_frames, int16_t* spectra, pdmp3_gc_side* side);
/* block until the slot's PCM is in its pinned buffer (no-op if nothing is in flight) */
int pdmp3_hip_stream_wait(pdmp3_hip_stream* hs, int slot);
/* 1: the wait above would return at once (nothing in flight on the slot, or the GPU is through with it), 0: not yet,
 * < 0: bad argument.  Never blocks. */
int pdmp3_hip_stream_done(pdmp3_hip_stream* hs, int slot);

/* Host-side twin of the generator (fills host buffers); used to build
 * identical inputs for the CPU baseline without a device round trip. */
int pdmp3_host_generate_frames(uint64_t seed, int64_t first_frame, int n_frames,
                               int16_t* spectra, pdmp3_gc_side* side);

#ifdef __cplusplus
}
#endif
#endif /* PDMP3_HIP_H */
