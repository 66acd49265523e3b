/* stream_api.c -- libpdmp3.so: the handle and the reference's streaming API (include/pdmp3.h; P:2351-2540): pdmp3_new /
 * delete / open_feed / feed / read / decode / getformat, pdmp3_amd_set_encoding / set_quirks; the read-ahead that turns the
 * frames the ring holds into one GPU batch, and the helper threads that share a batch's main data.
 * See host_internal.h for the map of the library. */
#include "host_internal.h"

/* ------------------------------------------------------------------------ */
/* handle                                                                    */
/* ------------------------------------------------------------------------ */

static pthread_mutex_t g_ctx_lock = PTHREAD_MUTEX_INITIALIZER;
#define MAX_DEVICES 16
static pdmp3_hip_ctx* g_ctx[MAX_DEVICES];

/* one engine (tables in HBM) per HIP device, shared by every handle / bulk decoder of the process on that device */
pdmp3_hip_ctx* shared_ctx_on(int dev) {
  if (dev < 0 || dev >= MAX_DEVICES) return NULL;
  pthread_mutex_lock(&g_ctx_lock);
  /* Several decoders = several HIP streams whose kernels should overlap; HIP multiplexes streams onto 4 hardware
   * queues by default and kernels of one queue run one after the other (measured, 4 decoders on the C4 corpus:
   * 5.5 M frames/s with 4 queues, 7.1 M with 16).  Only a default, and only effective if HIP is not up yet. */
  setenv("GPU_MAX_HW_QUEUES", "16", 0);
  if (!g_ctx[dev] && pdmp3_hip_create(dev, &g_ctx[dev]) != PDMP3_HIP_OK) g_ctx[dev] = NULL;
  pdmp3_hip_ctx* c = g_ctx[dev];
  pthread_mutex_unlock(&g_ctx_lock);
  return c;
}

int default_device(void) {
  const char* e = getenv("PDMP3_DEVICE");
  return e ? atoi(e) : 0;
}

static pdmp3_hip_ctx* shared_ctx(void) { return shared_ctx_on(default_device()); }

/* P:2351: pdmp3_new(decoder, error) -- `decoder` is ignored like in the reference */
pdmp3_handle* pdmp3_new(const char* decoder, int* error) {
  (void)decoder;
  pthread_once(&g_lut_once, build_luts);
  pdmp3_handle* id = (pdmp3_handle*)calloc(1, sizeof *id);
  if (!id) { if (error) *error = PDMP3_ERR; return NULL; }
  pdmp3_hip_ctx* ctx = shared_ctx();
  if (!ctx || pdmp3_hip_stream_create(ctx, BATCH_MAX, &id->hs) != PDMP3_HIP_OK) {
    fprintf(stderr, "pdmp3: no MI355X transform engine: %s\n", pdmp3_hip_last_error());
    free(id);
    if (error) *error = PDMP3_ERR;
    return NULL;
  }
  if (error) *error = PDMP3_OK;
  return id;
}

/* Test hook (host-logic tests on machines without a GPU): a handle that can
 * parse and tap records but not decode.  Not part of the reference API. */
pdmp3_handle* pdmp3_amd_new_parse_only(void) {
  pthread_once(&g_lut_once, build_luts);
  pdmp3_handle* id = (pdmp3_handle*)calloc(1, sizeof *id);
  if (id) id->host_only = 1;
  return id;
}

void pdmp3_amd_set_tap(pdmp3_handle* id, int16_t* spectra, pdmp3_gc_side* side, int cap_frames) {
  id->tap_spectra = spectra; id->tap_side = side; id->tap_cap = cap_frames; id->tap_n = 0;
}
int pdmp3_amd_tap_count(const pdmp3_handle* id) { return id->tap_n; }

static int ra_rollback(pdmp3_handle* id);      /* read-ahead of pdmp3_read, below */

/* P:2360 */
void pdmp3_delete(pdmp3_handle* id) {
  if (!id) return;
  if (id->ra_inflight && id->hs) (void)pdmp3_hip_stream_wait(id->hs, 0);
  if (id->hs) pdmp3_hip_stream_destroy(id->hs);
  free(id);
}

/* P:2369-2384 */
int pdmp3_open_feed(pdmp3_handle* id) {
  if (!id) return PDMP3_ERR;
  if (ra_rollback(id) != PDMP3_OK) return PDMP3_ERR;   /* (what survives open_feed is the parse state of the reference's position) */
  id->ostart = 0; id->istart = 0; id->iend = 0; id->processed = 0; id->new_header = 0;
  id->l_istart = 0; id->l_processed = 0; id->l_new_header = 0;
  id->need_reset = 1;
  id->main_top = 0;
  return PDMP3_OK;
}


/* P:2391-2423: all-or-nothing copy into the ring */
int pdmp3_feed(pdmp3_handle* id, const unsigned char* in, size_t size) {
  if (!(id && in && size)) return PDMP3_ERR;
  if (size > (size_t)ring_free_logical(id)) return PDMP3_NO_SPACE;
  size_t first;
  const int real = id->vsrc == NULL;             /* (virtual ring: the same index arithmetic, no bytes moved) */
  if (!real) id->vfed += size;
  if (id->iend < id->l_istart) {
    first = id->l_istart - id->iend;
    if (size < first) first = size;
    if (real) memcpy(id->in + id->iend, in, first);
    id->iend += (unsigned)first;
  } else {
    first = INBUF_SIZE - id->iend;
    if (size < first) first = size;
    if (first) { if (real) memcpy(id->in + id->iend, in, first); id->iend += (unsigned)first; size -= first; }
    if (size) { if (real) memcpy(id->in, in + first, size); id->iend = (unsigned)size; }
  }
  /* a feed that fills the ring exactly leaves iend == istart, which the reference reads as EMPTY (P:1062-1068): it
   * will not get to the frames read ahead before its next feeds have overwritten them */
  if (id->ra_head != id->ra_n && id->iend == id->l_istart && ra_rollback(id) != PDMP3_OK) return PDMP3_ERR;
  return PDMP3_OK;
}

/* ------------------------------------------------------------------------ */
/* Read-ahead.  pdmp3_read is synchronous per call (P:2431-2481): it parses   */
/* and decodes as many frames as the caller's buffer takes.  A GPU batch of   */
/* the three or four frames a 16 KiB buffer takes is all launch latency, so   */
/* when the parser has to parse a frame anyway it goes on through EVERY       */
/* complete frame the ring already holds and the engine decodes them as one   */
/* batch; later calls hand those frames out without parsing or launching.     */
/*                                                                            */
/* What the caller can observe must not change, so the handle keeps two       */
/* views of the stream: the parser's (istart, processed, hdr, new_header:     */
/* possibly ahead) and the reference's (l_*: what P:2431-2481 would have      */
/* consumed by now) -- pdmp3_feed's free space, the 1152-byte rule (H10),     */
/* pdmp3_getformat and the return codes use the second.  A frame is parsed    */
/* ahead only if the reference is certain to parse it to the same result      */
/* later: at least 1152 bytes are buffered behind its start NOW (more can     */
/* only be fed), the parse succeeded, and no step of it found the ring short  */
/* of bytes (ring_short) -- bytes fed later cannot change it then.  The       */
/* first frame of a batch is the one the reference parses in this very call:  */
/* its failures keep their side effects (H9: the reservoir keeps the bytes,   */
/* the ring is rewound); a frame read ahead that fails is undone completely   */
/* (snapshot) and left for the call in which the reference gets to it.        */
/* ------------------------------------------------------------------------ */
static void snap_save(const pdmp3_handle* id, parse_snap* p) {
  p->processed = id->processed; p->istart = id->istart; p->hdr = id->hdr; p->si = id->si;
  memcpy(p->scalefac_l, id->scalefac_l, sizeof p->scalefac_l);
  memcpy(p->scalefac_s, id->scalefac_s, sizeof p->scalefac_s);
  memcpy(p->count1, id->count1, sizeof p->count1);
  memcpy(p->main_vec, id->main_vec, sizeof p->main_vec); p->main_top = id->main_top;
  memcpy(p->side_vec, id->side_vec, sizeof p->side_vec); p->side_ptr = id->side_ptr; p->side_idx = id->side_idx;
  p->new_header = id->new_header; p->need_reset = id->need_reset; p->tap_n = id->tap_n;
}
static void snap_restore(pdmp3_handle* id, const parse_snap* p) {
  id->processed = p->processed; id->istart = p->istart; id->hdr = p->hdr; id->si = p->si;
  memcpy(id->scalefac_l, p->scalefac_l, sizeof p->scalefac_l);
  memcpy(id->scalefac_s, p->scalefac_s, sizeof p->scalefac_s);
  memcpy(id->count1, p->count1, sizeof p->count1);
  memcpy(id->main_vec, p->main_vec, sizeof p->main_vec); id->main_top = p->main_top;
  memcpy(id->side_vec, p->side_vec, sizeof p->side_vec); id->side_ptr = p->side_ptr; id->side_idx = p->side_idx;
  id->new_header = p->new_header; id->need_reset = p->need_reset; id->tap_n = p->tap_n;
}

static void ra_push(pdmp3_handle* id) {
  struct ra_entry* e = &id->ra[id->ra_n++];
  e->processed_after = id->processed; e->istart_after = id->istart; e->hdr = id->hdr;
  e->nch = (uint8_t)(id->hdr.mode == 3 ? 1 : 2);
  e->nh = (uint8_t)(id->new_header != 0);
}

/* Take back the frames read ahead but not handed out: the parser returns to the reference's position, the engine's
 * synthesis state to the last frame handed out.  Needed when the reference will NOT find what was read ahead: a
 * feed that fills the ring exactly makes it look empty to the reference (iend == istart, P:1062-1068), whose next
 * feeds then overwrite the unread frames; pdmp3_open_feed keeps the parse state of ITS position (H4-H6). */
static int ra_rollback(pdmp3_handle* id) {
  if (id->ra_head == id->ra_n) return PDMP3_OK;
  snap_restore(id, &id->ra_before[id->ra_head]);
  int rc = PDMP3_OK;
  if (id->hs && pdmp3_hip_stream_rewind(id->hs, 0, id->ra_head) != PDMP3_HIP_OK) {
    fprintf(stderr, "pdmp3: engine failure: %s\n", pdmp3_hip_last_error());
    rc = PDMP3_ERR;
  }
  id->ra_head = id->ra_n = id->ra_inflight = 0;
  return rc;
}

/* ------------------------------------------------------------------------ */
/* Helpers for a read-ahead batch.  What is sequential in a frame -- ring,    */
/* header, side info, bit reservoir -- is a fraction of a microsecond; its    */
/* main data (scalefactors + Huffman, 4-5 us at 320 kbps) only needs the      */
/* reservoir as that frame left it, and read_ahead has a copy of exactly that */
/* per frame (the snapshots it keeps for undoing frames).  So a batch's main  */
/* data is decoded by the caller AND a few helper threads, frame by frame off */
/* one counter; the merge into the state that survives frames (apply_main)    */
/* stays sequential.  The helpers are per process, started on first use, spin */
/* for a short while after a batch (the next one is usually 50-100 us away)   */
/* and then sleep.  PDMP3_STREAM_THREADS = helpers (default min(3, CPUs - 1); */
/* 0: none).  Handles that read at the same time SHARE them (round 6): each   */
/* batch goes into one of HP_SLOTS slots and a helper takes a frame from every */
/* slot that has one in turn; a handle that finds all slots taken decodes its  */
/* batch alone.                                                               */
/* ------------------------------------------------------------------------ */
#define HP_MAX 15
#define HP_SLOTS 4
typedef struct { const uint8_t* res; const frame_header* H; const side_info* S; main_out* out; } hp_job;
typedef struct {
  pthread_mutex_t own;             /* one batch at a time per slot */
  hp_job job[BATCH_MAX];
  _Alignas(64) _Atomic uint64_t state;   /* batch number << 32 | frames of the batch << 16 | next frame: ONE word, so that a
                                      helper that is late for a batch can never take a frame of it by the numbers of the next */
  _Atomic int done;
} hp_slot;
static struct {
  pthread_mutex_t start;           /* hp_start once */
  pthread_mutex_t m; pthread_cond_t cv;
  int started, n, sleepers;
  _Atomic uint32_t bell;           /* batches posted so far, all slots: what an idle helper watches */
  hp_slot slot[HP_SLOTS];
} g_hp = {.start = PTHREAD_MUTEX_INITIALIZER, .m = PTHREAD_MUTEX_INITIALIZER, .cv = PTHREAD_COND_INITIALIZER,
          .slot = {{.own = PTHREAD_MUTEX_INITIALIZER}, {.own = PTHREAD_MUTEX_INITIALIZER}, {.own = PTHREAD_MUTEX_INITIALIZER}, {.own = PTHREAD_MUTEX_INITIALIZER}}};
_Static_assert(HP_SLOTS == 4, "g_hp's initialiser names four slots");

/* Takes ONE frame of the slot's batch if it has one left.  Whatever the fetch-and-add returns IS a claim -- batch, frame
 * count and index come out of one word -- also for a thread that arrives here still thinking of the batch before: it must
 * decode the frame it drew, nobody else will.  (The look before the add keeps the index field from running over into the
 * count: an idle slot is looked at, not added to.) */
static inline int hp_take_one(hp_slot* sl) {
  uint64_t v = atomic_load_explicit(&sl->state, memory_order_acquire);
  if ((uint32_t)(v & 0xffff) >= (uint32_t)(v >> 16 & 0xffff)) return 0;
  v = atomic_fetch_add_explicit(&sl->state, 1, memory_order_acq_rel);
  const uint32_t k = (uint32_t)(v & 0xffff), n = (uint32_t)(v >> 16 & 0xffff);
  if (k >= n) return 0;
  const hp_job* j = &sl->job[k];
  decode_main(j->res, j->H, j->S, j->out);
  atomic_fetch_add_explicit(&sl->done, 1, memory_order_release);
  return 1;
}
/* How long a helper looks for the next batch before it sleeps on the condition: PDMP3_STREAM_SPIN = pause instructions
 * (default 1000: ~15 us; rounds 3-5: 20000, 0.2-0.5 ms -- three cores at 100 % per streaming handle; the
 * next batch of a caller that reads at the reference driver's cadence is 50-100 us away, measured with both:
 * profiles/r06_stream_api.json). */
static int g_hp_spin = 1000;
static void* hp_worker(void* arg) {
  const unsigned me = (unsigned)(uintptr_t)arg;
  uint32_t seen = 0;
  for (;;) {
    int spins = 0;
    while (atomic_load_explicit(&g_hp.bell, memory_order_acquire) == seen) {
      if (++spins < g_hp_spin) { hp_pause(); continue; }      /* a short look, then sleep */
      pthread_mutex_lock(&g_hp.m);
      g_hp.sleepers++;
      while (atomic_load_explicit(&g_hp.bell, memory_order_acquire) == seen) pthread_cond_wait(&g_hp.cv, &g_hp.m);
      g_hp.sleepers--;
      pthread_mutex_unlock(&g_hp.m);
      spins = 0;
    }
    seen = atomic_load_explicit(&g_hp.bell, memory_order_acquire);   /* (a batch posted from here on rings again) */
    for (int took = 1; took;) {                                      /* a frame from every slot that has one, in turn */
      took = 0;
      for (unsigned i = 0; i < HP_SLOTS; i++) took |= hp_take_one(&g_hp.slot[(me + i) % HP_SLOTS]);
    }
  }
  return NULL;
}
static void hp_start(void) {                                   /* (g_hp.start held) */
  const char* e = getenv("PDMP3_STREAM_THREADS");
  int n = e ? atoi(e) : usable_cpus() - 1;
  if (!e && n > 3) n = 3;
  if (n > HP_MAX) n = HP_MAX;
  const char* sp = getenv("PDMP3_STREAM_SPIN");
  if (sp && atoi(sp) >= 0) g_hp_spin = atoi(sp);
  for (int i = 0; i < n; i++) {
    pthread_t t;
    if (pthread_create(&t, NULL, hp_worker, (void*)(uintptr_t)i) != 0) break;
    pthread_detach(t);
    g_hp.n++;
  }
  __atomic_store_n(&g_hp.started, 1, __ATOMIC_RELEASE);
}
/* decode_main of jobs[0..n) */
static void hp_run(const hp_job* jobs, int n) {
  if (n > 1) {
    if (!__atomic_load_n(&g_hp.started, __ATOMIC_ACQUIRE)) {
      pthread_mutex_lock(&g_hp.start);
      if (!g_hp.started) hp_start();
      pthread_mutex_unlock(&g_hp.start);
    }
    hp_slot* sl = NULL;
    if (g_hp.n > 0)
      for (int i = 0; i < HP_SLOTS && !sl; i++) if (pthread_mutex_trylock(&g_hp.slot[i].own) == 0) sl = &g_hp.slot[i];
    if (sl) {
      memcpy(sl->job, jobs, (size_t)n * sizeof *jobs);
      atomic_store_explicit(&sl->done, 0, memory_order_relaxed);
      const uint32_t batch = (uint32_t)(atomic_load_explicit(&sl->state, memory_order_relaxed) >> 32) + 1;
      atomic_store_explicit(&sl->state, (uint64_t)batch << 32 | (uint64_t)n << 16, memory_order_release);
      atomic_fetch_add_explicit(&g_hp.bell, 1, memory_order_release);
      pthread_mutex_lock(&g_hp.m);
      if (g_hp.sleepers) pthread_cond_broadcast(&g_hp.cv);
      pthread_mutex_unlock(&g_hp.m);
      while (hp_take_one(sl)) { }                                /* the caller works on its OWN batch only */
      while (atomic_load_explicit(&sl->done, memory_order_acquire) < n) {   /* its last frames are in helpers' hands: a frame of
                                                                               another handle's batch meanwhile, if there is one */
        int took = 0;
        for (int i = 0; i < HP_SLOTS && !took; i++) if (&g_hp.slot[i] != sl) took = hp_take_one(&g_hp.slot[i]);
        if (!took) hp_pause();
      }
      pthread_mutex_unlock(&sl->own);
      return;
    }
  }
  for (int i = 0; i < n; i++) decode_main(jobs[i].res, jobs[i].H, jobs[i].S, jobs[i].out);
}

/* Parse the frame the reference parses now and, behind it, every frame that qualifies; send them to the engine.
 * Called with nothing read ahead (parser == logical view).  Returns the code of the FIRST frame's Read_Frame.
 * Three steps: everything that touches the ring, frame after frame; the frames' main data (hp_run); the merge
 * into the scalefactor / count1 state and the records, frame after frame. */
static int read_ahead(pdmp3_handle* id) {
  static _Thread_local int16_t scratch_sp[BATCH_MAX * 2304];      /* parse-only test handles: records go nowhere */
  static _Thread_local pdmp3_gc_side scratch_sd[BATCH_MAX * 4];
  int16_t* spectra = id->hs ? pdmp3_hip_stream_spectra(id->hs) : scratch_sp;
  pdmp3_gc_side* side = id->hs ? pdmp3_hip_stream_side(id->hs) : scratch_sd;
  id->ra_head = id->ra_n = 0;
  const size_t pos = id->processed;
  const unsigned mark = id->istart;
  snap_save(id, &id->ra_before[0]);
  const int res = read_frame_staged(id);
  if (res != PDMP3_OK) {                                  /* failed: rewind to the frame start (P:2459-2462) */
    id->processed = pos; id->istart = mark;
    sync_logical(id);                                     /* (the header it read stays, as in the reference) */
    return res;
  }
  const int reset0 = id->need_reset;
  id->need_reset = 0;
  ra_push(id);
  const int cap = getenv("PDMP3_NO_READAHEAD") ? 1 : BATCH_MAX;
  while (id->ra_n < cap) {
    /* bytes behind the parser, counted from the reference's cursor (a cursor that has crossed the end of the ring
     * while pdmp3_feed has iend parked there, P:2410-2417, sees the ring full of its own stale bytes) */
    const size_t ahead = id->processed - id->l_processed;
    const unsigned have = ring_filled_logical(id);
    const unsigned avail = have > ahead ? (unsigned)(have - ahead) : 0;
    if (avail < 1152) break;                              /* H10: the reference would not attempt it yet */
    parse_snap* snap = &id->ra_before[id->ra_n];
    snap_save(id, snap);
    id->ring_short = 0;
    const int r = read_frame_staged(id);
    /* undone unless it succeeded on bytes that were all there -- and belongs into this batch: the engine takes LSF
     * frames in launches of their own, all of one channel count (pdmp3_hip_stream_set_lsf) */
    if (r != PDMP3_OK || id->ring_short || id->processed - snap->processed > avail ||
        id->hdr.ver != id->ra[0].hdr.ver || (id->hdr.ver && (id->hdr.mode == 3) != (id->ra[0].hdr.mode == 3))) {
      snap_restore(id, snap);
      break;
    }
    ra_push(id);
  }
  /* frame i's header, side info and reservoir: what the snapshot taken before frame i + 1 holds -- the parser itself for the last */
  const int n = id->ra_n;
  hp_job jobs[BATCH_MAX];
  for (int i = 0; i < n; i++) {
    const parse_snap* nx = i + 1 < n ? &id->ra_before[i + 1] : NULL;
    jobs[i].res = nx ? nx->main_vec : id->main_vec;
    jobs[i].H = nx ? &nx->hdr : &id->hdr;
    jobs[i].S = nx ? &nx->si : &id->si;
    jobs[i].out = &id->ra_out[i];
    id->ra_out[i].is = spectra + (size_t)i * 2304;
  }
  hp_run(jobs, n);
  for (int i = 0; i < n; i++) {
    if (i) {                                              /* the snapshot before frame i gets the state the frames before it left */
      parse_snap* sn = &id->ra_before[i];
      memcpy(sn->scalefac_l, id->scalefac_l, sizeof sn->scalefac_l);
      memcpy(sn->scalefac_s, id->scalefac_s, sizeof sn->scalefac_s);
      memcpy(sn->count1, id->count1, sizeof sn->count1);
      sn->tap_n = id->tap_n;
    }
    apply_main(id, jobs[i].H, jobs[i].out);
    emit_records(id, jobs[i].H, jobs[i].S, i == 0 ? reset0 : 0, spectra + (size_t)i * 2304, side + (size_t)i * 4);
  }
  if (id->hs) {
    (void)pdmp3_hip_stream_set_lsf(id->hs, id->ra[0].hdr.ver != 0);
    if (pdmp3_hip_stream_submit(id->hs, 0, id->ra_n) != PDMP3_HIP_OK) {
      fprintf(stderr, "pdmp3: engine failure: %s\n", pdmp3_hip_last_error());
      id->ra_n = 0;
      return PDMP3_ERR;
    }
    id->ra_inflight = 1;
  }
  return res;
}

/* pdmp3_read (P:2431-2481) */
static int read_impl(pdmp3_handle* id, unsigned char* outmemory, size_t outsize, size_t* done) {
  *done = 0;
  int res = PDMP3_ERR;
  if (id->ostart) {                               /* rest of the frame a previous call could not fit */
    const size_t n = drain_frame(id, outmemory, outsize);
    *done = n; outsize -= n; outmemory += n;
    res = PDMP3_OK;
  }
  while (outsize) {
    if (id->ra_head == id->ra_n) {                /* nothing read ahead: the reference's own step */
      if (ring_filled_logical(id) < 1152) { res = PDMP3_NEED_MORE; break; }      /* H10 */
      res = read_ahead(id);
      if (id->ra_n == 0) {
        if (res == PDMP3_OK || res == PDMP3_NEW_FORMAT) return PDMP3_ERR;        /* engine failure */
        break;
      }
    } else {
      if (ring_filled_logical(id) < 1152) {       /* (cannot be after a plain feed; the net under ra_rollback's cases) */
        if (ra_rollback(id) != PDMP3_OK) return PDMP3_ERR;
        res = PDMP3_NEED_MORE;
        break;
      }
      res = PDMP3_OK;                             /* Read_Frame of a frame read ahead: it succeeded */
    }
    /* Decode_L3 + Convert_Frame_S16 of the frame at the head */
    const struct ra_entry* e = &id->ra[id->ra_head];
    if (!id->hs && !id->host_only) return PDMP3_ERR;
    if (id->ra_inflight) {
      if (pdmp3_hip_stream_wait(id->hs, 0) != PDMP3_HIP_OK) {
        fprintf(stderr, "pdmp3: engine failure: %s\n", pdmp3_hip_last_error());
        return PDMP3_ERR;
      }
      id->ra_inflight = 0;
    }
    id->l_processed = e->processed_after; id->l_istart = e->istart_after; id->l_hdr = e->hdr;
    if (!id->l_new_header && e->nh) id->l_new_header = 1;
    /* where frame ra_head of the batch lies in the slot's PCM (include/pdmp3_hip.h): an MPEG-1 frame in its own 4608-byte
     * place (9216 as float); LSF frames -- half the samples -- back to back when stereo, in pairs per place when mono */
    const size_t place = id->enc_f32 ? 9216u : 4608u;
    const size_t fbytes = (e->hdr.ver ? place / 4 : place / 2) * e->nch;
    const size_t off = !e->hdr.ver ? (size_t)id->ra_head * place
                       : e->nch == 2 ? (size_t)id->ra_head * fbytes : (size_t)(id->ra_head >> 1) * place + (size_t)(id->ra_head & 1) * fbytes;
    const unsigned char* pcm = id->hs ? (const unsigned char*)pdmp3_hip_stream_pcm(id->hs) + off : NULL;
    id->ra_head++;
    if (pcm && id->ostart == 0 && outsize >= fbytes) {    /* whole frame fits: copy straight through */
      memcpy(outmemory, pcm, fbytes);
      outmemory += fbytes; outsize -= fbytes; *done += fbytes;
      id->last_nch = e->nch;
    } else {
      /* Convert_Frame_S16 (P:2307-2345) starts at the cursor it finds: a new frame does not reset it.  (The cursor
       * is not 0 here only after a call whose buffer ended inside a sample-frame -- 1..3 stray bytes -- which makes
       * the reference decode and drop frames; the new frame is then handed out from that sample on.) */
      if (pcm) memcpy(id->last_pcm, pcm, fbytes);
      else memset(id->last_pcm, 0, fbytes);       /* parse-only test handle: silence */
      id->last_nch = e->nch;
      const size_t n = drain_frame(id, outmemory, outsize);
      outmemory += n; outsize -= n; *done += n;
    }
  }
  if (id->l_new_header == 1 && res == PDMP3_OK) res = PDMP3_NEW_FORMAT;
  return res;
}

int pdmp3_read(pdmp3_handle* id, unsigned char* outmemory, size_t outsize, size_t* done) {
  if (!(id && outmemory && outsize && done)) return PDMP3_ERR;
  return read_impl(id, outmemory, outsize, done);
}

/* parse-only variant of the read loop for host-logic tests: parses every frame
 * the ring allows, taps records, produces no PCM */
int pdmp3_amd_parse_available(pdmp3_handle* id) {
  int16_t sp[2304];
  pdmp3_gc_side sd[4];
  int res = PDMP3_NEED_MORE;
  while (ring_filled(id) >= 1152) {
    const size_t pos = id->processed;
    const unsigned mark = id->istart;
    res = read_frame(id, sp);
    if (res != PDMP3_OK && res != PDMP3_NEW_FORMAT) { id->processed = pos; id->istart = mark; sync_logical(id); return res; }
    emit_records(id, &id->hdr, &id->si, id->need_reset, sp, sd);
    id->need_reset = 0;
    sync_logical(id);
  }
  return PDMP3_NEED_MORE;
}

/* P:2491-2520 */
int pdmp3_decode(pdmp3_handle* id, const unsigned char* in, size_t insize, unsigned char* out, size_t outsize, size_t* done) {
  size_t take = ring_free_logical(id);
  *done = 0;
  if (take > insize) take = insize;               /* the surplus is silently dropped (H16) */
  int res = pdmp3_feed(id, in, take);
  if (res != PDMP3_OK) return res;
  if (out && outsize) {
    size_t got;
    res = pdmp3_read(id, out, outsize, &got);
    *done = got;
  } else if (id->l_processed == 0) {              /* probe: peek at the first header, then rewind */
    const size_t pos = id->processed;             /* (nothing is read ahead before the first frame is handed out) */
    const unsigned mark = id->istart;
    res = search_header(id);
    id->processed = pos; id->istart = mark;
    sync_logical(id);
    if (id->l_new_header == 1) res = PDMP3_NEW_FORMAT;
  }
  return res;
}

/* include/pdmp3.h: float output (not in the reference) */
int pdmp3_amd_set_encoding(pdmp3_handle* id, int encoding) {
  if (!id || (encoding != PDMP3_ENC_SIGNED_16 && encoding != PDMP3_ENC_FLOAT_32)) return PDMP3_ERR;
  const int want = encoding == PDMP3_ENC_FLOAT_32;
  if (want == id->enc_f32) return PDMP3_OK;
  if (ra_rollback(id) != PDMP3_OK) return PDMP3_ERR;          /* frames read ahead were decoded in the other format */
  if (id->ostart) {                                            /* the frame under the cursor was, too: convert what is left */
    const unsigned nch = id->l_hdr.mode == 3 ? 1 : 2;
    const unsigned spf = frame_samples(&id->l_hdr);
    if (want) { float* f = (float*)id->last_pcm; for (int k = (int)(spf * nch) - 1; k >= 0; k--) f[k] = (float)id->last_pcm[k] / 32767.0f; }
    else { const float* f = (const float*)id->last_pcm; for (unsigned k = 0; k < spf * nch; k++) { float v = f[k] * 32767.0f; id->last_pcm[k] = (int16_t)(v > 32767.0f ? 32767 : v < -32767.0f ? -32767 : (int)v); } }
  }
  if (id->hs && pdmp3_hip_stream_set_f32(id->hs, want) != PDMP3_HIP_OK) return PDMP3_ERR;
  id->enc_f32 = want;
  return PDMP3_OK;
}

/* P:2526-2535 */
/* ISO-correct switches (include/pdmp3.h): from the next frame parsed on.  Frames that pdmp3_read has
 * parsed ahead keep the mode they were parsed in. */
int pdmp3_amd_set_quirks(pdmp3_handle* id, unsigned iso_mask) {
  if (!id || (iso_mask & ~(PDMP3_ISO_ALL | PDMP3_ISO_LSF))) return PDMP3_ERR;
  id->iso = iso_mask;
  return PDMP3_OK;
}

int pdmp3_getformat(pdmp3_handle* id, long* rate, int* channels, int* encoding) {
  if (!(id && rate && channels && encoding)) return PDMP3_ERR;
  *encoding = id->enc_f32 ? PDMP3_ENC_FLOAT_32 : PDMP3_ENC_SIGNED_16;
  *rate = (long)kLsfSampleRates[sfreq9(&id->l_hdr)];
  *channels = id->l_hdr.mode == 3 ? 1 : 2;
  id->new_header = -1;
  id->l_new_header = -1;
  return PDMP3_OK;
}

