/*
 * pdmp3.h -- the libmpg123-style streaming API of technosaurus/PDMP3
 * (pdmp3.c:115-157), implemented by pdmp3_amd/libpdmp3.so with the Layer-III
 * transforms running on an MI355X through include/pdmp3_hip.h.
 *
 * Same names, argument meaning, return codes and error behaviour as the
 * reference so that its `main.c` (and any libmpg123-feed-style caller) links
 * against this library unchanged:
 *
 *   pdmp3_new        pdmp3.c:2351     pdmp3_delete     pdmp3.c:2360
 *   pdmp3_open_feed  pdmp3.c:2369     pdmp3_feed       pdmp3.c:2391
 *   pdmp3_read       pdmp3.c:2431     pdmp3_decode     pdmp3.c:2491
 *   pdmp3_getformat  pdmp3.c:2526     pdmp3            pdmp3.c:2540
 *
 * Differences by design:
 *   1. the handle is opaque (the reference publishes its struct, pdmp3.c:124-148, but no caller in its tree or of
 *      its API touches a field; this library's handle holds pinned staging, queue state and device objects);
 *   2. there is no PDMP3_HEADER_ONLY mode (pdmp3.c:161: `#define PDMP3_HEADER_ONLY` + `#include "pdmp3.c"` to get
 *      the declarations without the code): this header plays that role, the code is a shared library;
 *   3. the Huffman table index `hufftables g_huffman_main[34]`, which the reference exports by accident
 *      (pdmp3.c:535, missing `static`), is not exported: the library's code books are derived tables with another
 *      layout (pdmp3_amd/csrc/tables_data.h);
 *   4. synthesis state is per handle instead of process-global; the handle starts zeroed (H13);
 *   5. additions, all prefixed pdmp3_amd_: float output below, whole-stream decoding in pdmp3_bulk.h.
 * Not a difference in results, but visible to a process: pdmp3_read decodes the frames the ring already holds as one
 * batch and uses helper threads for their scalefactors + Huffman data (started on first use, shared by all handles).
 * After a batch the helpers look for the next one for about 15 us (PDMP3_STREAM_SPIN pause instructions, default 1000: 101 k / 202 k frames/s at 25 / 13 us of CPU per frame against 109 k / 208 k at 37 / 19 with 20000, profiles/r06_stream_api.json;
 * 20000 = the 0.2-0.5 ms of rounds 3-5, which kept three more cores at 100 % under a caller that reads at the reference
 * driver's cadence) and then sleep on a condition.  Handles that read at the same time share them: each batch goes into one
 * of four slots and a helper takes a frame from every slot that has one in turn (a handle that finds all four taken decodes
 * its batch alone); three handles reading at once run within 3 % of each other (profiles/r06_stream_share.json).
 * PDMP3_STREAM_THREADS=0 turns the helpers off.
 * Environment: PDMP3_STREAM_THREADS = number of helpers (default min(3, CPUs - 1); 0 = the
 * calling thread only), PDMP3_STREAM_SPIN (above), PDMP3_NO_READAHEAD = one frame per batch, PDMP3_DEVICE = HIP device of
 * new handles.
 * There is no CPU decode path: pdmp3_new() returns NULL (and sets *error when given) if no HIP device / engine
 * library is available.
 */
#ifndef PDMP3_H
#define PDMP3_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PDMP3_OK           0
#define PDMP3_ERR         -1
#define PDMP3_NEED_MORE  -10
#define PDMP3_NEW_FORMAT -11
#define PDMP3_NO_SPACE     7

#define PDMP3_ENC_SIGNED_16 (0x080|0x040|0x10)

typedef struct pdmp3_handle pdmp3_handle;

pdmp3_handle* pdmp3_new(const char* decoder, int* error);
void pdmp3_delete(pdmp3_handle* id);
int pdmp3_open_feed(pdmp3_handle* id);
int pdmp3_feed(pdmp3_handle* id, const unsigned char* in, size_t size);
int pdmp3_read(pdmp3_handle* id, unsigned char* outmemory, size_t outsize, size_t* done);
int pdmp3_decode(pdmp3_handle* id, const unsigned char* in, size_t insize,
                 unsigned char* out, size_t outsize, size_t* done);
int pdmp3_getformat(pdmp3_handle* id, long* rate, int* channels, int* encoding);

/* Output encoding (NOT in the reference, which has int16 only).  PDMP3_ENC_FLOAT_32 -- libmpg123's
 * value for 32-bit float -- makes pdmp3_read / pdmp3_decode deliver interleaved native float, full scale +-1.0,
 * unclipped: the binary32 synthesis sum that pdmp3.c:2028-2031 would scale by 32767, truncate and clip, so the
 * int16 output is exactly clip(trunc(float * 32767)).  Sizes count bytes as before (8 per stereo sample-frame).
 * Takes effect at once; frames decoded but not handed out yet are decoded again.  pdmp3_getformat reports it.
 * Returns PDMP3_ERR for any other encoding. */
#define PDMP3_ENC_FLOAT_32 0x200
int pdmp3_amd_set_encoding(pdmp3_handle* id, int encoding);

/* ISO-correct switches (NOT in the reference).  The reference departs from ISO 11172-3 in the places
 * below (SURVEY H1-H5, and what an independent decoder showed on top of H3 in round 6) and this library reproduces them
 * by default -- that is what "drop-in" means.  A bit set here selects the standard's behaviour for that item instead,
 * from the next frame parsed on:
 *   PDMP3_ISO_TABLE33   H1  count1table_select = 1 decodes with the standard's table B (4-bit codes); the reference's
 *                           table index points into the middle of table 24 (pdmp3.c:569)
 *   PDMP3_ISO_MS_BOUND  H2  MS stereo on EVERY line (pdmp3.c:1920 stops at the smaller of the two channels' count1 --
 *                           and counts in reordered lines, so that in short blocks even "the larger count1" would
 *                           leave coded lines unrotated)
 *   PDMP3_ISO_IS_SHORT  H3  intensity stereo on short blocks multiplies by the ratios, window by window
 *                           (pdmp3.c:2191 keeps them in `unsigned`, 2212-2213 assign the sample to both channels)
 *   PDMP3_ISO_SF21      H4  long scalefactor band 21 has scalefactor 0 (pdmp3.c:1896-1901 reads scalefac_l[21], the next
 *                           array's first element)
 *   PDMP3_ISO_SF12      H5  short scalefactor band 12 has scalefactor 0 (pdmp3.c:1864-1869 reads scalefac_s[12][w], for
 *                           granule 1 / channel 1 the bits of the previous granule's output)
 *   PDMP3_ISO_IS_BOUND      the rest of the standard's intensity stereo (2.4.3.4.9.3): is_pos is the RIGHT channel's
 *                           scalefactor (pdmp3.c:2163, 2200 read the left one's), the intensity region starts above the
 *                           right channel's last non-zero line -- per window in short blocks -- (pdmp3.c:1946-1965: above
 *                           its count1), the last band takes the position of the band below it (pdmp3.c:1953, 1961 leave
 *                           it out), intensity-coded lines are not M/S-rotated.  Implies PDMP3_ISO_IS_SHORT's arithmetic.
 * Nothing in the reference defines these modes.  Since round 6 they are PINNED by an independent ISO decoder: FFmpeg's
 * mpegaudiodec decodes the packer's conforming streams to within 2 LSB of PDMP3_ISO_ALL (tests/golden/iso_*.npz,
 * tools/make_iso_golden.py).  Returns PDMP3_ERR for unknown bits. */
#define PDMP3_ISO_TABLE33  0x01u
#define PDMP3_ISO_MS_BOUND 0x02u
#define PDMP3_ISO_IS_SHORT 0x04u
#define PDMP3_ISO_SF21     0x08u
#define PDMP3_ISO_SF12     0x10u
#define PDMP3_ISO_IS_BOUND 0x20u
#define PDMP3_ISO_ALL      0x3fu
/* ... and one more bit that is not about HOW MPEG-1 is decoded but about WHAT is accepted (not part of PDMP3_ISO_ALL):
 *   PDMP3_ISO_LSF       MPEG-2 LSF (22.05 / 24 / 16 kHz) and "MPEG-2.5" (11.025 / 12 / 8 kHz) Layer III streams (ISO/IEC
 *                       13818-3) are decoded -- one granule, 576 sample-frames per frame.  The reference returns an error
 *                       for their headers (pdmp3.c:1293) and so does this library without the bit.  An LSF frame is decoded
 *                       by the standard throughout (the other switches are implied for it: there is no reference behaviour
 *                       to reproduce); pinned by FFmpeg's decode of packer streams of all six rates
 *                       (tests/golden/lsf_*.npz).  pdmp3_read hands out frames of 576 sample-frames then; its "1152 bytes
 *                       buffered" rule is unchanged.  The whole-stream decoder takes LSF streams with its
 *                       Huffman stage on the host (include/pdmp3_bulk.h). */
#define PDMP3_ISO_LSF      0x40u
int pdmp3_amd_set_quirks(pdmp3_handle* id, unsigned iso_mask);

/* CLI driver: NULL-terminated list of .mp3 paths ("-" = stdin); writes
 * <first file>.raw (interleaved native-endian int16), as the reference's
 * OUTPUT_RAW build does (pdmp3.c:2236-2257).  A leading "/dev/dsp*" argument
 * is accepted and ignored (OSS playback is out of scope). */
void pdmp3(char* const* mp3s);

#ifdef __cplusplus
}
#endif
#endif
