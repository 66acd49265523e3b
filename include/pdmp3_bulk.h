/*
 * pdmp3_bulk.h -- whole-stream decoding on top of include/pdmp3.h
 * (pdmp3_amd/libpdmp3.so).
 *
 * NOT part of the reference's API: the reference decodes one frame per
 * pdmp3_read() iteration on one core.  This is the throughput form of the same
 * contract for callers that hold a whole file in memory (first
 * "next" row: the host Huffman stage that feeds the transforms).  The OUTPUT
 * is defined by the reference: byte for byte what its CLI driver pdmp3()
 * (pdmp3.c:2540-2589: 4096-byte feeds, 16 KiB reads, raw int16 sink) writes
 * for the same bytes, including the dropped tail and the stop at
 * the first frame error.
 *
 * Inside: the reference's read loop runs sequentially but only as far as the
 * bit reservoir (everything that depends on the previous frame); scalefactor +
 * Huffman decoding of each frame, a pure function of its reservoir snapshot,
 * runs on a pool of host threads straight into pinned staging memory; windows
 * of frames go through pdmp3_hip_stream_submit() (include/pdmp3_hip.h) three
 * deep, so upload, transforms, download and host decoding overlap.
 */
#ifndef PDMP3_BULK_H
#define PDMP3_BULK_H

#include <stddef.h>
#include <stdint.h>
#include "pdmp3_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bulk pdmp3_amd_bulk;

/* Returned by the scan / decode / parse entry points when the stream drives the reference into replaying its
 * 16 KiB input ring: pdmp3_feed leaves the write index AT the end of the ring when a feed ends exactly there
 * (pdmp3.c:2410-2417), and a frame that then ends exactly there too -- only 1152-byte frames can: 32 kHz at
 * 256 kbps, the limit of SURVEY H10 -- wraps the read index past it (pdmp3.c:1464-1474), after which the ring
 * looks full of its own stale bytes.  The reference's CLI then writes the same audio again, usually forever
 * (libpdmp3.so's pdmp3_read / pdmp3() keep that behaviour, as a drop-in must).  There is no finite reference
 * output to reproduce, so the whole-stream entry points stop and say so. */
#define PDMP3_BULK_REPLAY (-2)

/* threads <= 0: one worker per usable CPU (affinity and cgroup quota; at most 64; 4 with device Huffman, where the
 * pool only copies PCM out).  window_frames <= 0: with device Huffman the engine's slots hold 8192 frames -- the split scan
 * (below) fills them as far as its scanners have got when a slot is free, the one-thread scan closes its windows at 4096 --,
 * with host Huffman 2048; a size given explicitly is both (at most 32768).  Returns NULL when there is no transform
 * engine (no CPU fallback). */
pdmp3_amd_bulk* pdmp3_amd_bulk_new(int threads, int window_frames);
/* Environment: PDMP3_BULK_HOST_HUFFMAN=1 (below), PDMP3_BULK_SNAPSHOT_ROWS=1 (upload 2064-byte reservoir snapshots per frame
 * instead of the compact pool + row descriptors: tests), PDMP3_BULK_TRACE=1 (one summary of the pipeline's waits per decode on
 * stderr; 2: a line per window as well), PDMP3_BULK_SCAN_THREADS=n (0 .. 16: scanner threads of the split scan -- a pre-pass
 * hops from header to header and n threads run the scan from every 256th frame into private windows that the calling
 * thread puts into the engine's slots in stream order, results identical to the one-thread scan; by default 12 / 8 / 4 / 2 with
 * 32 / 16 / 12 / 6 usable CPUs when the PCM stays in device memory, where the scan is the bound, four of them at most for
 * destinations in host memory (there with the engine's windows closed at 4096 frames; decoders made with windows shorter than
 * 1024 frames, and hosts with fewer than 12 usable CPUs, keep the one-thread scan); given explicitly: that many for every destination and for streams from 4 private
 * windows on instead of 12; 0: never; decoders of one process that scan
 * at the same time share: the second takes half the scanners, the third a third ...), PDMP3_BULK_PREPASS_THREADS=n (1 .. 8
 * parts of the pre-pass, all but the first with a thread of their own that hops from a guessed header; 6 / 3 / 1 by default
 * with 16 / 12 / fewer usable CPUs; no part shorter than 256 KB), PDMP3_BULK_SUB_FRAMES=n (frames of a private window; 256 by
 * default), PDMP3_BULK_SCAN_SPIN=0 / 1 (a scanner whose snapshot is the next or the one after yields / spins while it waits; by
 * default it spins where the process has 2 x scanners + 8 CPUs),
 * PDMP3_BULK_GATHER_THREADS=n (0 .. 8 helper threads for the copies of the windows' main data into the pinned upload
 * buffers; 6 by default with 8 scanners, 3 with fewer, 0 without), PDMP3_BULK_GATHER_NT=0 (those copies with memcpy instead
 * of non-temporal stores).
 * Thread footprint of ONE decoder with a device destination on a host with 16 usable CPUs or more, for its lifetime
 * (the threads are started when a stream first needs them and then sleep on a job queue between streams): 8 scanners,
 * 5 hop threads + the pre-pass, 6 gather helpers, the submitter and the copy-out pool (`threads`) -- about 25.  Only
 * the scanners are shared out between decoders of one process that scan at the same time (above); gather helpers,
 * hop threads and the pre-pass are per decoder, and their waits yield the CPU (sched_yield) or nap 20 us at a time rather
 * than sleep on a condition (the scanners' wait for their snapshots: a broadcast per snapshot to a dozen sleepers was
 * the split scan's bottleneck until round 5).  A
 * process that keeps many decoders (one per GPU and more) should size them with the variables above --
 * PDMP3_BULK_SCAN_THREADS=2 PDMP3_BULK_PREPASS_THREADS=1 PDMP3_BULK_GATHER_THREADS=1 is 6 threads per decoder. */
/* host_huffman = 0 (what pdmp3_amd_bulk_new gives unless PDMP3_BULK_HOST_HUFFMAN=1 is set): the host only runs
 * the sequential scan and ships side info + reservoir snapshots; scalefactors, Huffman and the frame-to-frame
 * merge run on the device (pdmp3_hip_stream_submit_bits) and the pool just copies PCM out.  host_huffman = 1:
 * they run on the pool and the engine is given decoded records (pdmp3_hip_stream_submit). */
pdmp3_amd_bulk* pdmp3_amd_bulk_new_ex(int threads, int window_frames, int host_huffman);
/* the same on HIP device `device` (the others use $PDMP3_DEVICE, default 0).  Decoders are independent: a corpus of
 * files is dealt over the GPUs of a node by giving each host thread its own decoder (whole files per GPU,
 * largest first; pdmp3_amd/sharding.py assign_files) -- there is nothing to exchange between them. */
pdmp3_amd_bulk* pdmp3_amd_bulk_new_on(int threads, int window_frames, int host_huffman, int device);
void pdmp3_amd_bulk_delete(pdmp3_amd_bulk* b);
int pdmp3_amd_bulk_threads(const pdmp3_amd_bulk* b);
/* Streams the decoder's split scan (several scanner threads; taken for device destinations, or with
 * PDMP3_BULK_SCAN_THREADS) decoded to their end, and streams it gave up half way -- irregular ones: resync, tags, a
 * truncation the ring's feed cadence shows -- and decoded again with the one-thread scan.  Same PCM either way. */
void pdmp3_amd_bulk_split_scans(const pdmp3_amd_bulk* b, long long* taken, long long* given_up);
/* ISO-correct switches (include/pdmp3.h: PDMP3_ISO_*, pdmp3_amd_set_quirks) for the streams decoded from now on;
 * 0 (the default) = the reference's behaviour, bit for bit.  The CLI driver reads the mask from $PDMP3_CLI_ISO. */
int pdmp3_amd_bulk_set_quirks(pdmp3_amd_bulk* b, unsigned iso_mask);
/* PDMP3_ISO_LSF (MPEG-2 LSF / MPEG-2.5 streams, which the reference rejects): the device's Huffman stage reads MPEG-1 side
 * info only, so LSF streams take the HOST Huffman stage -- a host_huffman decoder decodes them in its windows like any
 * stream (a window closes where the version, or an LSF stream's channel count, changes); a device-Huffman decoder
 * hands a stream that holds an LSF frame ANYWHERE to a host-Huffman decoder it creates for the purpose (same device,
 * threads, window and switches; the call is synchronous then): at once when the stream opens with an LSF header, else when
 * its scan meets the first one (what had gone to the GPU by then is dropped, the stream is decoded again from its first
 * byte) -- what counts as a frame never depends on which stage decodes the Huffman data. */

/* PCM bytes (return value) and frames pdmp3() would produce for this stream;
 * header / side-info / reservoir pass only, no Huffman, no GPU.  Use it to
 * size the output of pdmp3_amd_bulk_decode. */
long long pdmp3_amd_scan_buffer(const unsigned char* mp3, size_t n, long long* frames);
/* ... with a decoder's switches (pdmp3_amd_bulk_set_quirks): PDMP3_ISO_LSF makes MPEG-2 LSF / 2.5 frames count */
long long pdmp3_amd_scan_buffer_iso(const unsigned char* mp3, size_t n, unsigned iso_mask, long long* frames);

/* Decode one whole stream with a fresh decoder state.  Returns the PCM byte
 * count pdmp3() writes for it; pcm[0 .. min(return, pcm_cap)) holds them
 * (interleaved int16; bytes of `pcm` past the return value are unspecified).
 * -1 on an engine failure.  rate / channels (may be NULL): format of the last
 * header seen, as pdmp3_getformat reports it. */
long long pdmp3_amd_bulk_decode(pdmp3_amd_bulk* b, const unsigned char* mp3, size_t n,
                                unsigned char* pcm, size_t pcm_cap, long* rate, int* channels);

/* Optional: PCM buffers in pinned host memory.  When the `pcm` given to the decode calls lies in such a buffer the
 * GPU downloads every window straight into it (mono frames packed on the way) and the host never touches the
 * samples; any other memory works too, through a staging buffer and a copy by the pool.  A DEVICE pointer (hipMalloc,
 * a torch tensor) is accepted as `pcm` as well: the PCM then never leaves the GPU -- the decode kernel stores it there
 * itself when the destination is memory of the decoder's device and takes whole frames (copies otherwise).  The decoder
 * works on HIP streams of its own, which do NOT wait for the caller's: whatever the caller still has in flight on the
 * buffer (a fill, an earlier consumer) must be complete before the decode call, and a consumer on another stream starts
 * after pdmp3_amd_bulk_decode / pdmp3_amd_bulk_wait has returned. */
void* pdmp3_amd_pcm_alloc(size_t bytes);
void pdmp3_amd_pcm_free(void* p);

/* The same without waiting for the tail: returns as soon as the stream is scanned and its windows are queued on
 * the device (`mp3` may be released then, `pcm` must stay); the PCM of every stream given so far is complete after
 * pdmp3_amd_bulk_wait() returns 0.  The next stream's scan overlaps the previous one's GPU work and copy-out, so a
 * corpus of medium-sized files runs at the long-stream rate instead of paying a pipeline fill and drain per file.
 * Each stream starts from a fresh decoder state, exactly as with pdmp3_amd_bulk_decode (PDMP3_FR_RESET /
 * PDMP3_FR_NEWSTREAM on its first frame; nothing is reset from the host).  Host-Huffman decoders simply wait. */
long long pdmp3_amd_bulk_decode_async(pdmp3_amd_bulk* b, const unsigned char* mp3, size_t n,
                                      unsigned char* pcm, size_t pcm_cap, long* rate, int* channels);
int pdmp3_amd_bulk_wait(pdmp3_amd_bulk* b);

/* Host stages only, for tests on machines without a GPU: a decoder made by
 * pdmp3_amd_bulk_new_parse_only() writes the gc records the engine would be
 * given into caller memory (cap_frames frames of 2304 int16 / 4 records).
 * Returns the frame count, -1 if they do not fit. */
pdmp3_amd_bulk* pdmp3_amd_bulk_new_parse_only(int threads, int window_frames);
long long pdmp3_amd_bulk_parse(pdmp3_amd_bulk* b, const unsigned char* mp3, size_t n,
                               int16_t* spectra, pdmp3_gc_side* side, size_t cap_frames, long long* pcm_bytes);

/* The scan alone, in the device-Huffman form: per frame the pdmp3_frame_bits and PDMP3_RESERVOIR_BYTES of
 * reservoir that the engine would be given (host tests). */
pdmp3_amd_bulk* pdmp3_amd_bulk_new_parse_bits(void);
long long pdmp3_amd_bulk_parse_bits(pdmp3_amd_bulk* b, const unsigned char* mp3, size_t n, pdmp3_frame_bits* bits,
                                    uint8_t* reservoir, size_t cap_frames, long long* pcm_bytes);

/* The same in the compact form the engine is actually given (include/pdmp3_hip.h, pdmp3_row_desc): side info, row
 * descriptors and the pool, as ONE window holding the whole stream (host tests: the rows rebuilt from it must be the
 * snapshots of pdmp3_amd_bulk_parse_bits).  pool_cap: capacity of `pool`; *pool_bytes: what was used. */
long long pdmp3_amd_bulk_parse_pool(pdmp3_amd_bulk* b, const unsigned char* mp3, size_t n, pdmp3_frame_bits* bits,
                                    pdmp3_row_desc* desc, uint8_t* pool, size_t pool_cap, size_t cap_frames, size_t* pool_bytes);

/* The streaming API (include/pdmp3.h) driven from a memory buffer by a C loop: pdmp3_new, pdmp3_open_feed, then
 * pdmp3_read(read_bytes) until PDMP3_ERR, with a pdmp3_feed of feed_bytes on every PDMP3_NEED_MORE -- the
 * reference driver's loop (pdmp3.c:2564-2584; 4096 and 16384 there) with its two sizes as parameters.  eager != 0:
 * the caller keeps the ring as full as feed_bytes-sized feeds allow instead of waiting for PDMP3_NEED_MORE, but never
 * to the last byte (a ring filled exactly looks EMPTY to the reference, pdmp3.c:1062-1068, and so to this library).  At most
 * `cap` bytes go to `out` (may be NULL); returns the PCM bytes the reads delivered, -1 without an engine.  For
 * measuring the drop-in path without an interpreter in the loop (bench.py: streaming_api). */
long long pdmp3_amd_stream_loop(const unsigned char* mp3, size_t n, unsigned char* out, size_t cap,
                                size_t feed_bytes, size_t read_bytes, int eager);

/* A PCM buffer as a RIFF/WAVE file: interleaved int16 (float32 = 0) or 32-bit float (float32 != 0; the output of
 * pdmp3_amd_set_encoding(PDMP3_ENC_FLOAT_32)).  The reference's only sink is the raw writer (pdmp3.c:2236-2257); the
 * CLI driver writes "<first name>.wav" instead of ".raw" when PDMP3_CLI_WAV=1 is set.  PDMP3_OK or PDMP3_ERR. */
int pdmp3_amd_write_wav(const char* path, const void* pcm, size_t bytes, long rate, int channels, int float32);

/* A corpus of whole files over the GPUs of a node -- SURVEY 8e's second partitioning ("C4: whole files per GPU, largest
 * first"), the C form of pdmp3_amd/sharding.py assign_files + one decoder per device (nothing of the
 * reference's: it has one handle and one thread).  devices[]: HIP device numbers, one host thread and one whole-stream
 * decoder per entry (a device listed twice gets two decoders); the files are dealt largest first, each device's files are
 * decoded one after the other with the asynchronous call, nothing is exchanged between devices.  pcm[i] (capacity
 * pcm_caps[i]; size it with pdmp3_amd_scan_buffer_iso) receives file i's PCM, pcm_bytes[i] the byte count pdmp3() would
 * write for it.  threads_per_decoder / window_frames / host_huffman as for pdmp3_amd_bulk_new_on.  0, or -1. */
void pdmp3_amd_corpus_assign(const size_t* sizes, int n_files, int world, int* rank_of);
int pdmp3_amd_corpus_decode(const int* devices, int n_devices, const unsigned char* const* mp3s, const size_t* sizes, int n_files,
                            unsigned char* const* pcm, const size_t* pcm_caps, long long* pcm_bytes, unsigned iso_mask,
                            int threads_per_decoder, int window_frames, int host_huffman);

#ifdef __cplusplus
}
#endif
#endif
