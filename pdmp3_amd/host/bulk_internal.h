/*
 * bulk_internal.h -- the whole-stream decoder's types (include/pdmp3_bulk.h is its interface): a decoder (struct bulk),
 * its windows and flights, the split scan's records and snapshots; shared by bulk.c, split_scan.c, bulk_api.c, corpus.c
 * and wav_cli.c.  Nothing here is exported.
 */
#ifndef PDMP3_BULK_INTERNAL_H
#define PDMP3_BULK_INTERNAL_H
#include "host_internal.h"
#include "../../include/pdmp3_bulk.h"
#define BULK_SLOTS 6
#define BULK_GATH_EXTRA 16             /* copy-list entries beyond one per frame: segment images of a split scan's windows */
#define GATHER_MAX_HELPERS 8
#define GATHER_QUEUE 512
#define GATHER_TASK_ENTRIES 512          /* copy-list entries per task: half a megabyte of main data */
#define PAR_MAX_BATCH 64               /* private windows of a split scan that go into one window of the engine, at most */
#include <time.h>
static inline double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }
#define PDMP3_BULK_REPLAY (-2)         /* see bulk_drive */
#define BULK_GRAB 8                   /* frames a worker takes per trip to the counter */
#define BULK_COPY_PIECE ((size_t)256 << 10)

typedef struct {
  frame_header hdr;
  side_info si;
  uint8_t reset;
  uint8_t res[RESERVOIR_BYTES];
} frame_job;

typedef struct {
  frame_job* jobs;
  main_out* outs;
  int n;
  int16_t* spectra;                   /* destination of this window's records */
  pdmp3_gc_side* side;
  int slot;
} bulk_window;

typedef struct {                      /* a window that is on the GPU */
  int n, active, all_stereo;          /* all_stereo: 2 = every frame stereo, 1 = every frame mono, 0 = mixed */
  unsigned char* dst;                 /* where this window's PCM goes (caller memory) and how much room is left there */
  size_t dst_cap;
  int direct;                         /* the GPU downloads straight to dst (pinned caller memory): nothing to copy */
  uint8_t* nch;
  long long sub_seq;                  /* its place in the submitter's queue */
  int lsf;                            /* a window of LSF frames (all of one version and channel count): half the PCM per frame,
                                         laid out as include/pdmp3_hip.h pdmp3_hip_decode_lsf_frames says */
} bulk_flight;

/* split scan: a window as a scanner thread leaves it -- what bits_push / fill_reservoir_pool write into an engine slot,
 * in private memory.  The bytes the scanner itself puts into the pool (a segment's image of the reservoir buffer) are
 * kept in `arena` and entered in the copy list like the main data the submitter gathers from the stream. */
#define PW_ARENA_BYTES (24u << 10)
struct par_cache;
typedef struct pre_window {
  long long index;                    /* window number within the stream */
  int cap;                            /* frames its arrays hold */
  struct par_cache* home;             /* where it goes when the stitcher is through with it (NULL: freed) */
  int n, gath_n;
  size_t pool_tail;
  pdmp3_frame_bits* bits;
  pdmp3_row_desc* desc;
  uint8_t* nch;
  void* gath;                         /* struct pool_copy[] */
  uint8_t* arena;
  size_t arena_len;
  double t_take, t_begin, t_done;     /* trace: taken by a scanner, its snapshot there, pushed */
} pre_window;
struct par_scan;

struct bulk {
  pdmp3_handle* id;
  int cap;                            /* frames a window holds */
  int target;                         /* frames the one-thread scan gives a window (<= cap) */
  int ramp_on;                        /* this stream's first windows are short (win_ramp): long streams only */
  int cur_target;                     /* this stream's window size: the slot's capacity for a stream that fits one slot, else `target` */
  int trace2; double tr_t0;           /* $PDMP3_BULK_TRACE >= 2: per-window lines, times from the stream's start */
  int count_only;                     /* scan: stage A alone */
  int bits_mode;                      /* main data goes to the device undecoded (pdmp3_hip_stream_submit_bits) */
  int device, window_arg;             /* what bulk_new was given (the LSF decoder below is made with the same) */
  struct bulk* lsf_alt;               /* bits mode + PDMP3_ISO_LSF: the host-Huffman decoder LSF streams go through (bulk_decode_impl) */
  pdmp3_frame_bits* bits_dst; uint8_t* res_dst;   /* where stage A writes the current window (bits mode) */
  int bits_n, bits_slot, bits_open;
  pdmp3_frame_bits* rec_bits; uint8_t* rec_res;   /* parse-only bits mode: caller memory */
  /* compact bits input (include/pdmp3_hip.h: pdmp3_row_desc): res_dst is the window's pool */
  int pool_mode;                      /* 1: rows go up as pool + descriptors; 0: as 2064-byte snapshots */
  pdmp3_row_desc* desc_dst;
  size_t pool_tail, pool_cap;
  int seg_first;                      /* index in the window of the first frame of the current segment */
  uint32_t seg_s_off;
  int need_segment;                   /* id->main_vec is the live buffer: the next regular frame starts a segment */
  uint32_t cur_row_off; unsigned cur_top; int cur_explicit, cur_staged;   /* what the frame just staged leaves for its descriptor */
  /* The frames' main data is not copied by the scanning thread (it is memory-bound there: two thirds of its time per
   * frame): stage A only notes where the bytes are in the caller's stream and where they go in the pool, and the
   * submitter thread copies them just before the window goes up (pool_gather).  What the scanner itself needs from
   * the pool before that -- a few KB per window, pool_materialize -- it copies early (pool_ensure). */
  struct pool_copy { const uint8_t* src; uint32_t dst, n; } *gath[BULK_SLOTS], *gath_cur;
  int gath_n, gath_cap;                           /* entries written / entries gath_cur has room for */
  cpu_set_t near_gpu;                             /* the CPUs of the GPU's NUMA node this process may use (empty: no binding) */
  int sky[RESERVOIR_BYTES + 1], sky_n;            /* frames of the segment no later frame has topped yet (pdmp3_row_desc.up) */
  pdmp3_row_desc* rec_desc; size_t rec_pool_cap;  /* parse-only pool mode (host tests): caller memory, one window */
  bulk_window win[2];
  int cur;                            /* window stage A is filling */
  bulk_window* in_b;                  /* window the workers hold, or NULL */
  long long windows;                  /* windows handed to the workers so far */
  long long frames;
  /* workers */
  pthread_t* th;
  int nth;
  pthread_mutex_t mu;
  pthread_cond_t cv_work, cv_done;
  bulk_window* work;
  const unsigned char* copy_src; unsigned char* copy_dst; size_t copy_bytes, copy_next, copy_row;
  long long gen;
  int next, active, quit;
  /* sinks: the engine (hs) or caller memory (parse only, host tests) */
  pdmp3_hip_stream* hs;
  bulk_flight flight[BULK_SLOTS];
  int16_t* rec_spectra; pdmp3_gc_side* rec_side; size_t rec_cap;
  unsigned char* pcm; size_t pcm_cap;
  size_t pcm_emitted;                 /* PCM bytes of all frames handed to stage C so far */
  int pcm_pinned;                     /* the caller's PCM buffer is pinned host memory (1) or device memory (2): direct target */
  int failed, busy;
  int carry;                          /* keep parse state (host handle / device sfstate) from the previous stream */
  /* bits mode: the engine calls of a window (H2D, kernels, D2H: ~40 us of driver time) are issued by a thread
   * of their own, in window order, so that the scanning thread does not stop for them */
  pthread_t sub_th;
  int sub_started, sub_quit, sub_rc;
  pthread_mutex_t sub_mu;
  pthread_cond_t sub_cv, sub_done_cv;
  int sub_slot[8], sub_n[8], sub_row[8];
  size_t sub_pool[8];                 /* pool bytes of the window (0: snapshot rows) */
  int sub_gath[8];                    /* entries of the slot's copy list */
  void* sub_dst[8];
  long long sub_head, sub_tail;       /* jobs enqueued / completed */
  long long sub_copied;               /* jobs whose main data is out of the caller's stream (<= sub_tail + 1) */
  long long par_taken, par_given_up;  /* streams the split scan took to their end / gave up half way (pdmp3_amd_bulk_split_scans) */
  size_t next_copy_row;               /* row size of the copy job bulk_collect last handed out */
  /* split scan (par_scan below): a scanner that fills private windows (struct pre_window) instead of engine slots */
  int scan_threads;                   /* scanners of the split scan (0: stage A on the calling thread alone) */
  int scan_forced;                    /* PDMP3_BULK_SCAN_THREADS was given: split whatever the destination */
  long long stream_win;               /* windows of the CURRENT stream closed so far (the first ones are short: win_frames) */
  /* the submitter's helpers: a window's main data (1 KB per frame, out of the caller's stream into the slot's pinned pool)
   * is copied by several threads at once -- on one thread it is 9 ms of an hour of audio, which is what bounds the
   * pipeline once the scan is split */
  /* the main-data copies of the windows (pool_gather), as tasks: helper threads take them, and so does whoever waits for
   * a slot's copies to be complete.  The one-thread scan's windows are cut into tasks when the submitter gets them; the
   * split scan's as its private windows are put into the slot, so that the copies are under way while the window fills. */
  pthread_t gh_th[GATHER_MAX_HELPERS];
  int gh_n, gh_quit;
  struct { uint8_t* pool; const void* list; int n, slot; } gq[GATHER_QUEUE];
  unsigned gq_head, gq_tail;          /* pushed / taken */
  int g_out[BULK_SLOTS];              /* tasks of the slot not finished yet */
  int g_pushed[BULK_SLOTS];           /* the open window's copies have been handed out as it was filled */
  pthread_mutex_t gh_mu; pthread_cond_t gh_cv, gh_done_cv;
  int win_sink;
  struct pre_window* pw_cur;
  struct par_scan* par;               /* where finished windows go */
  struct par_cache* pc;               /* the split scan's memory, kept from stream to stream */
  long long limit_frames;             /* stop parsing when `frames` reaches this (0: no limit) */
  double tr_take, tr_begin;           /* trace stamps for the window being scanned */
  uint8_t* priv_pool;                 /* win_sink: this scanner's pool of the window it is filling */
  void* slot_arena[BULK_SLOTS][PAR_MAX_BATCH];   /* the literal bytes (segment images) a slot's copy list points into, from its pre_windows */
  int slot_arena_n[BULK_SLOTS];
  double t_submit, t_gpuwait, t_poolwait;   /* PDMP3_BULK_TRACE=1: where the scanning thread waits */
  double t_sub_gather, t_sub_call, t_drive, t_subwait, t_tail;          /* ... and what the submitter thread spends on the main-data copies / the engine calls */
};

/* room for a segment start (2064 + 511), a frame's main data (< 2000) and an explicit image (2064) */
#define POOL_ROOM 6700u
_Static_assert(POOL_ROOM <= PDMP3_POOL_SLACK_BYTES, "a fresh window has room for its first frame");
static inline int bulk_at_limit(const struct bulk* b) { return (b->limit_frames && b->frames >= b->limit_frames) || b->id->lsf_seen; }

typedef struct {              /* what a frame's header and side info say by themselves (hop_parse) */
  uint32_t x;
  uint16_t fb, begin, top;    /* frame bytes, main_data_begin, begin + main-data bytes */
  uint8_t nch, crc;
  uint8_t ws;                 /* win_switch_flag of granule-channel g = gr * 2 + ch in bit g */
  uint8_t pad;
} hop1;
/* The pre-pass is itself a chain -- where a header is follows from the one before -- but only from a known header on:
 * HOP threads start at guessed places (the stream cut into equal parts), look for a header there that three more
 * headers follow, and hop from it; the pre-pass proper walks the first part itself and from there on reads the hop
 * threads' records instead of the stream (4 ns a frame instead of 27).  A guess is right when the part before it lands
 * exactly on it; one that is not (a header-like pattern inside main data that chains three times) makes the stream go
 * the one-thread way, as everything else the pre-pass does not like does. */
enum { SEG_RUNNING = 0, SEG_AT_NEXT, SEG_AT_END, SEG_BAD };
typedef struct pre_seg {
  struct par_scan* P; int j;
  size_t guess;
  hop1* rec; long long cap;
  long long x_start;          /* where its first header is: -2 not known yet, -1 none found (atomic) */
  char pad0[64];              /* (what the hop thread keeps writing has cache lines of its own: its neighbours write theirs as often) */
  long long count;            /* records written (atomic, release; moved on every 8 frames and at the end) */
  int state;                  /* SEG_* (atomic, release; final once not SEG_RUNNING) */
  double t_sync, t_done;      /* trace */
  char pad1[64];
} pre_seg;
#define PAR_MAX_SCANNERS 16
#define PAR_MAX_SEGS 8
#define PAR_AHEAD 64
/* What the split scan allocates per stream is tens of megabytes in blocks large enough for malloc to map and unmap each
 * time: every page of them faults in again on every stream, in threads that share one address space (3 ms of a 6 ms
 * decode on a 256-core host).  The decoder keeps them instead: the records of the pre-pass, each scanner's scratch, the
 * private windows (as many as can be in flight). */
struct hop_rec_s;
typedef struct par_cache {
  pthread_mutex_t mu;
  struct hop_rec_s* rec; long long rec_cap;
  hop1* seg_rec[PAR_MAX_SEGS]; long long seg_cap[PAR_MAX_SEGS];
  struct { struct bulk* wb; pdmp3_handle* id; uint8_t* pool; size_t pool_bytes; } scan[PAR_MAX_SCANNERS];
  pre_window* spare[2 * PAR_AHEAD]; int n_spare;
  /* the threads of the scan (hop threads, pre-pass, scanners): started when a stream first needs them, kept for the next
   * stream -- fourteen pthread_create / pthread_join per stream are 0.2 ms, as much as the scan of a five-minute file */
  pthread_mutex_t crew_mu; pthread_cond_t crew_cv, crew_done_cv;
  pthread_t crew[PAR_MAX_SCANNERS + PAR_MAX_SEGS];
  int crew_n, crew_busy, crew_quit;               /* threads / jobs taken or waiting to be taken */
  cpu_set_t near_gpu;                             /* where they run (struct bulk::near_gpu; empty: anywhere) */
  struct { void* (*fn)(void*); void* arg; int* left; } jobs[PAR_MAX_SCANNERS + PAR_MAX_SEGS];
  unsigned job_head, job_tail;
} par_cache;
typedef struct hop_rec_s {
  uint32_t x;                 /* offset of the frame's header in the stream */
  uint32_t md_src;            /* offset of its main data */
  uint64_t md_end;            /* main-data bytes of the stream up to and including this frame */
  uint16_t fb, begin, top;    /* frame bytes, main_data_begin, reservoir fill after it */
  uint8_t nch, crc;
} hop_rec;

typedef struct {              /* the scanner's state in front of frame `frame` (a window boundary) */
  long long frame;
  unsigned istart, iend;
  size_t processed, vfed, fed;
  unsigned main_top;
  int sky_n;
  int* sky;                   /* frames whose bytes are still visible in the reservoir buffer, oldest (largest top) first */
  long long last_ws0[4], last_ws1[4];   /* the last frame before it whose gc g had win_switch_flag 0 / 1 (-1: none) */
  int ready;
} span_snap;

struct par_scan {
  /* ---- set before the threads start, read by all of them (the pre-pass looks at abort / quit once per frame: none of this
   *      shares a cache line with what the scanners and the stitcher write) */
  struct bulk* b;
  const unsigned char* mp3; size_t n;
  int K;
  /* Scanners take WINDOWS in turn (a shared counter): window w is scanned from the snapshot the pre-pass leaves at its
   * first frame.  (Contiguous spans per scanner starve the GPU while the first scanner works through its span alone: one
   * scanner produces windows at half the rate the GPU takes them.  Window by window, w is ready at
   * w x [pre-pass time per window] + [scan time of one window], always ahead of the GPU's w x 157 us.) */
  int sub;                    /* frames of a private window (the engine's windows are made of several: par_drive) */
  int spin;                   /* a scanner whose snapshot is the next or the one after spins for it (hosts with cores to spare); else it yields */
  int one_window;             /* by its first frame's size the stream fits one window of the engine */
  span_snap* snap; long long snap_cap;   /* by window index; [0] unused (a fresh handle) */
  hop_rec* rec; long long rec_cap;
  pre_window** win; long long win_cap;    /* finished windows by stream index (entries: under the mutex) */
  int J;                      /* parts of the pre-pass: [0] is the pre-pass thread's own, the others have a hop thread each */
  struct scanner_arg* args;
  double t0;
  atomic_int abort;           /* (set under the mutex, so that waiters wake; also looked at in loops that hold no lock) */
  atomic_int quit;            /* the stitcher has left: nobody wants further windows (not an error) */
  char pad0[64];
  /* ---- the pre-pass's: how far it is.  The scanners do not sleep on a condition for their snapshots -- with a dozen of them
   *      waiting, every broadcast (a snapshot, a finished window, a window taken) woke them all and the mutex they then queued
   *      for was the pre-pass's too: 2.7 ms of pre-pass with 8 scanners, 5.8 ms with 16 -- they watch this counter */
  long long published;        /* windows < this have their snapshot (atomic, release; 1 from the start: window 0 needs none) */
  long long n_frames;         /* valid once prepass_done */
  int prepass_done, irregular;   /* (prepass_done: atomic, release; set under the mutex as well: the stitcher sleeps on the condition) */
  double t_prepass, t_pre_part0, t_pre_wait;
  char pad1[64];
  /* ---- the scanners' and the stitcher's */
  long long next_win;         /* the next window nobody has taken yet (atomic) */
  long long stitched;         /* windows the stitcher has taken (atomic; written under the mutex) */
  int scanners_done;
  pthread_mutex_t mu; pthread_cond_t cv;
  int jobs_left, hops_left;   /* pre-pass and scanners / hop threads that have not returned (under the crew's mutex) */
  char pad2[64];
  pre_seg seg[PAR_MAX_SEGS];
};
#define PAR_NOT_TAKEN (-3)
#define PAR_GIVEN_UP (-4)

/* bulk.c: the worker pool, window rotation, the submitter and its copy tasks, the windows of the device Huffman path,
 * the one-thread scan (bulk_drive) and the read loop it runs (read_impl_sink) */
HOST_LOCAL void* bulk_worker(void* arg);
HOST_LOCAL void bulk_start_b(struct bulk* b, bulk_window* w, const unsigned char* src, unsigned char* dst, size_t nbytes);
HOST_LOCAL void bulk_wait_b(struct bulk* b);
HOST_LOCAL int bulk_collect(struct bulk* b, int slot, const unsigned char** jsrc, unsigned char** jdst, size_t* jbytes);
HOST_LOCAL int bulk_finish_b(struct bulk* b);
HOST_LOCAL int bulk_rotate(struct bulk* b);
HOST_LOCAL void pool_gather(uint8_t* pool, const struct pool_copy* g, int n);
HOST_LOCAL void* gather_helper(void* arg);
HOST_LOCAL void gather_push(struct bulk* b, int slot, uint8_t* pool, const struct pool_copy* g, int n);
HOST_LOCAL void* bulk_submitter(void* arg);
HOST_LOCAL int sub_drain_copied(struct bulk* b);
HOST_LOCAL int sub_drain(struct bulk* b);
HOST_LOCAL int bits_open_window(struct bulk* b);
HOST_LOCAL void pool_materialize(struct bulk* b);
HOST_LOCAL int bits_close_window(struct bulk* b);
HOST_LOCAL long long bulk_drive(struct bulk* b, const unsigned char* mp3, size_t n);
HOST_LOCAL int read_impl_sink(pdmp3_handle* id, size_t outsize, size_t* done, struct bulk* sink);
/* split_scan.c: private windows and the memory kept from stream to stream; par_drive returns PAR_NOT_TAKEN / PAR_GIVEN_UP
 * when the stream has to go the one-thread way */
HOST_LOCAL pre_window* pw_new_in(par_cache* pc, int cap, long long index);
HOST_LOCAL void pw_free(pre_window* w);
HOST_LOCAL int pw_close_window(struct bulk* b);
HOST_LOCAL void pc_free(par_cache* pc);
HOST_LOCAL long long par_drive(struct bulk* b, const unsigned char* mp3, size_t n, int K);
HOST_LOCAL void header_fields(uint32_t h, frame_header* H);
/* cpus.c */
HOST_LOCAL int gpu_local_cpus(pdmp3_hip_ctx* ctx, cpu_set_t* out);
HOST_LOCAL void bind_thread(pthread_t t, const cpu_set_t* set);

#endif
