/*
 * pdmp3_node.h -- one stream decoded by the GPUs of ONE node: the sharded form of the hot path behind a C-ABI.
 *
 * No reference counterpart (technosaurus/PDMP3 has no parallelism at all: pdmp3.c is one thread and one handle at a
 * time).  What it implements is SURVEY 8e: a stream's frames are cut into contiguous ranges, one per
 * GPU; the only coupling between neighbouring ranges is the synthesis history -- the IMDCT overlap (pdmp3.c:1775-1776)
 * and the polyphase FIFO (pdmp3.c:2006-2019), two granules deep -- so every range but the first starts two frames
 * early and discards what those frames decode to (further back where the frames in front of the cut are mono: channel
 * 1's state is the last stereo frame's), NO collective runs inside the decode, and the one exchange of the path is the
 * final PCM gather to the first GPU: ncclSend / ncclRecv over RCCL, every sender straight into its place in the
 * destination (xGMI is point to point: the root's seven ingress links work in parallel).  Since round 6 the exchange is
 * PIPELINED with the decode: a shard is decoded in slices (state carried from slice to slice, no extra halos) and a
 * slice's PCM leaves on a second stream while the next slices decode; every entry point puts the caller's current HIP
 * device back before it returns.
 *
 * In libpdmp3_hip.so (pdmp3_amd/csrc/node.hip).  RCCL is looked up at run time (librccl.so.1 / librccl.so, the copy a
 * process has loaded already if there is one): the library does not link against it, and a process that never creates
 * a node never loads it.  Plain C: pointers and sizes only.
 *
 * bench.py --gpus N does the same thing across N PROCESSES through torch.distributed (one rank per GPU, the contract of
 * this repository's driver); this is the form a C host program links against -- one process, one host thread per GPU.
 */
#ifndef PDMP3_NODE_H
#define PDMP3_NODE_H

#include <stddef.h>
#include <stdint.h>
#include "pdmp3_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pdmp3_node pdmp3_node;

/* how the PCM gets to the first device */
#define PDMP3_NODE_RCCL 0   /* ncclSend / ncclRecv, the root's receives grouped per slice (ncclCommInitAll over the devices): the
                               intended product path -- EXPERIMENTAL with more than one GPU: no machine this library was developed
                               on had two, the exchange between different devices has never executed (one rank: tested) */
#define PDMP3_NODE_COPY 1   /* device-to-device copies (hipMemcpyPeerAsync), no RCCL: the transport of the tests that run several
                               ranks on ONE GPU (RCCL refuses a device that is listed twice); same shards, same halos, same result */

typedef struct pdmp3_node_timing {
  double prepare_ms;        /* upload of the records (decode_records) / generation on the devices (decode_generated) */
  double decode_ms;         /* the kernels: first launch to the last one's end on a rank's compute stream (HIP events), the slowest rank */
  double gather_ms;         /* the exchange: first transfer's start to the last one's end on a rank's transfer stream, the slowest
                               rank -- it STARTS when the first slice is decoded and runs under the later slices' kernels */
  long long gather_bytes;   /* bytes that crossed between devices */
  int rccl_ranks;           /* ncclCommCount of rank 0's communicator (0 with PDMP3_NODE_COPY) */
  int slices;               /* pieces each shard was decoded and sent in ($PDMP3_NODE_SLICES, default 8; never under 4096 frames) */
  double total_ms;          /* decode + exchange as they ran, overlapped: wall clock from the first launch to the last byte */
} pdmp3_node_timing;

/* One engine (pdmp3_hip_create) and one HIP stream per entry of devices[]; with PDMP3_NODE_RCCL one communicator over
 * them.  A device may be listed more than once only with PDMP3_NODE_COPY.  PDMP3_HIP_OK or a PDMP3_HIP_E* code
 * (pdmp3_hip_last_error() has the text: no RCCL library, ncclCommInitAll refused, ...). */
int pdmp3_node_create(const int* devices, int n_devices, int transport, pdmp3_node** out);
void pdmp3_node_destroy(pdmp3_node* node);
int pdmp3_node_ranks(const pdmp3_node* node);

/* SURVEY 8e's partitioning, as the decode entry points apply it (== pdmp3_amd/sharding.py shard_with_halo, which the
 * N-process form uses): rank's share of n_frames frames is [lo, hi) with sizes that differ by at most one; it decodes
 * from *first (lo - 2, clamped; further back past a run of mono frames when frame_flags -- one pdmp3_gc_side.frame
 * byte per frame -- says so) and discards the PCM of the first *discard = lo - *first frames.  frame_flags may be NULL. */
void pdmp3_node_shard(long long n_frames, int rank, int world, const uint8_t* frame_flags,
                      long long* first, long long* count, long long* discard);

/* A stream that is at hand as records in HOST memory (spectra: n_frames x 2304 int16, side: n_frames x 4 records, the
 * layout of pdmp3_hip_decode_frames) -> n_frames x 4608 bytes of PCM in d_pcm, DEVICE memory on devices[0].  Every rank
 * uploads its own shard, decodes it from the zero state (rank 0: as pdmp3_hip_decode_frames with d_state = NULL) and
 * the shards' PCM is gathered.  A mono frame's PCM is the first 2304 bytes of its 4608-byte place, the rest of the place
 * is zero.  Synchronous: d_pcm is complete on return.  timing may be NULL. */
int pdmp3_node_decode_records(pdmp3_node* node, const int16_t* spectra, const pdmp3_gc_side* side, long long n_frames,
                              int16_t* d_pcm, pdmp3_node_timing* timing);

/* BASELINE configs[4]: the synthetic stream of pdmp3_hip_generate_frames(seed), n_frames frames; every
 * rank generates its shard on its own device (the generator is counter-based), decodes and the PCM is gathered as above. */
int pdmp3_node_decode_generated(pdmp3_node* node, uint64_t seed, long long n_frames, int16_t* d_pcm, pdmp3_node_timing* timing);

#ifdef __cplusplus
}
#endif
#endif
