/* bulk.c -- libpdmp3.so: the whole-stream decoder's pipeline on one scanning thread (include/pdmp3_bulk.h): the read loop
 * with a sink, windows and slots, the worker pool (host Huffman, PCM copies), the submitter thread and the main-data
 * copy tasks, the compact pool + descriptor upload of the device Huffman path.
 * See host_internal.h for the map of the library. */
#include "bulk_internal.h"

/* ------------------------------------------------------------------------ */
/* Bulk decode of one whole in-memory stream (first "next" row:   */
/* the host Huffman stage in front of the transforms).  Not a reference      */
/* entry point; its OUTPUT is defined by one: byte for byte what pdmp3()     */
/* (P:2540-2589) writes for the same file.                                   */
/*                                                                          */
/*   A  sequential   the reference's read loop at the CLI's cadence (4096 B  */
/*                   feeds, 16 KiB reads): ring, header sync, side info, bit */
/*                   reservoir -- everything whose result depends on the     */
/*                   previous frame.  Each frame leaves a job: header, side  */
/*                   info and a snapshot of the reservoir.                   */
/*   B  parallel     decode_main() per job on the worker threads, spectra    */
/*                   written straight into the engine's pinned staging slot. */
/*   C  sequential   apply_main() + emit_records() in frame order: the       */
/*                   scalefactor / count1 state that survives frames.        */
/*   D  GPU, async   pdmp3_hip_stream_submit(); the PCM of window w-2 is     */
/*                   copied out while w-1 is on the GPU and w is in B.       */
/* ------------------------------------------------------------------------ */

/* PCM out of a pinned slot into the caller's (pageable) memory.  The destination is written once and not read here
 * again, the source was written by the DMA engine and is in no cache: on x86-64 the copy goes through non-temporal
 * stores -- no read-for-ownership of the destination lines, a third less memory traffic per byte than memcpy below
 * glibc's own non-temporal threshold (tens of megabytes; the pieces here are 256 KB) -- with the source prefetched a
 * few lines ahead.  With four copy threads a window's 19 MB have to leave at 11 GB/s per thread to keep up with
 * PCIe; a host whose cores do 7 GB/s with plain memcpy was the 6.4 M frames/s of the round-4 driver run (the same
 * build did 9.7 M elsewhere).  PDMP3_BULK_PLAIN_COPY=1: memcpy. */
#if defined(__x86_64__)
#include <emmintrin.h>
static int g_plain_copy = -1;
static void copy_out(unsigned char* dst, const unsigned char* src, size_t n) {
  int plain = __atomic_load_n(&g_plain_copy, __ATOMIC_RELAXED);
  if (plain < 0) { const char* e = getenv("PDMP3_BULK_PLAIN_COPY"); plain = (e && *e == '1') ? 1 : 0; __atomic_store_n(&g_plain_copy, plain, __ATOMIC_RELAXED); }
  if (plain || n < 4096) { memcpy(dst, src, n); return; }
  const size_t head = (size_t)(-(uintptr_t)dst & 63);          /* up to the destination's next cache line */
  if (head) { memcpy(dst, src, head); dst += head; src += head; n -= head; }
  size_t lines = n >> 6;
  while (lines--) {
    _mm_prefetch((const char*)src + 512, _MM_HINT_NTA);
    const __m128i a = _mm_loadu_si128((const __m128i*)src), b = _mm_loadu_si128((const __m128i*)src + 1);
    const __m128i c = _mm_loadu_si128((const __m128i*)src + 2), d = _mm_loadu_si128((const __m128i*)src + 3);
    _mm_stream_si128((__m128i*)dst, a); _mm_stream_si128((__m128i*)dst + 1, b);
    _mm_stream_si128((__m128i*)dst + 2, c); _mm_stream_si128((__m128i*)dst + 3, d);
    src += 64; dst += 64;
  }
  _mm_sfence();
  if (n & 63) memcpy(dst, src, n & 63);
}
#else
static void copy_out(unsigned char* dst, const unsigned char* src, size_t n) { memcpy(dst, src, n); }
#endif

struct bulk;
static int bulk_push(struct bulk* b);             /* snapshot the frame read_frame_staged just staged */
static int bulk_at_limit(const struct bulk* b);
/* The whole-stream decoder's form of the read loop (`sink`): the call only does what touches the input ring and the
 * output cursor; main data decoding, the transforms and the PCM copy are the sink's business (bulk path below),
 * byte counts are the same.  The parser is never ahead here. */
int read_impl_sink(pdmp3_handle* id, size_t outsize, size_t* done, struct bulk* sink) {
  *done = 0;
  int res = PDMP3_ERR;
  if (id->ostart) {                               /* rest of the frame a previous call could not fit */
    const size_t n = drain_frame(id, NULL, outsize);
    *done = n; outsize -= n;
    res = PDMP3_OK;
  }
  while (outsize) {
    if (ring_filled(id) < 1152) { res = PDMP3_NEED_MORE; break; }      /* H10 */
    if (bulk_at_limit(sink)) { res = PDMP3_OK; break; }                /* (split scan: this scanner's span ends here) */
    const size_t pos = id->processed;
    const unsigned mark = id->istart;
    res = read_frame_staged(id);
    if (res != PDMP3_OK && res != PDMP3_NEW_FORMAT) {   /* failed: rewind to the frame start (P:2459-2462) */
      id->processed = pos; id->istart = mark;
      sync_logical(id);
      break;
    }
    sync_logical(id);
    if (bulk_push(sink) != PDMP3_OK) return PDMP3_ERR;
    id->last_nch = id->hdr.mode == 3 ? 1 : 2;
    const size_t n = drain_frame(id, NULL, outsize);      /* (the cursor is NOT reset for a new frame: P:2307-2345) */
    outsize -= n; *done += n;
  }
  if (id->l_new_header == 1 && res == PDMP3_OK) res = PDMP3_NEW_FORMAT;
  return res;
}



void* bulk_worker(void* arg) {
  struct bulk* b = (struct bulk*)arg;
  long long seen = 0;
  for (;;) {
    pthread_mutex_lock(&b->mu);
    while (b->gen == seen && !b->quit) pthread_cond_wait(&b->cv_work, &b->mu);
    if (b->quit) { pthread_mutex_unlock(&b->mu); return NULL; }
    seen = b->gen;
    bulk_window* w = b->work;
    const unsigned char* csrc = b->copy_src;
    unsigned char* cdst = b->copy_dst;
    const size_t cbytes = b->copy_bytes, crow = b->copy_row;
    pthread_mutex_unlock(&b->mu);
    while (w) {
      const int i0 = __atomic_fetch_add(&b->next, BULK_GRAB, __ATOMIC_RELAXED);
      if (i0 >= w->n) break;
      const int i1 = i0 + BULK_GRAB < w->n ? i0 + BULK_GRAB : w->n;
      for (int i = i0; i < i1; i++) {
        main_out* o = &w->outs[i];
        o->is = w->spectra + (size_t)i * 2304;
        decode_main(w->jobs[i].res, &w->jobs[i].hdr, &w->jobs[i].si, o);
      }
    }
    for (;;) {                                     /* PCM of an older window: pinned slot -> caller memory */
      const size_t c0 = __atomic_fetch_add(&b->copy_next, BULK_COPY_PIECE, __ATOMIC_RELAXED);
      if (c0 >= cbytes) break;
      const size_t c1 = cbytes - c0 < BULK_COPY_PIECE ? cbytes : c0 + BULK_COPY_PIECE;
      if (crow == 4608) copy_out(cdst + c0, csrc + c0, c1 - c0);
      else {                                       /* mono: 2304-byte frames out of 4608-byte slots */
        for (size_t off = c0; off < c1;) {
          const size_t r = off / 2304, w = off % 2304;
          const size_t n = 2304 - w < c1 - off ? 2304 - w : c1 - off;
          memcpy(cdst + off, csrc + r * 4608 + w, n);
          off += n;
        }
      }
    }
    pthread_mutex_lock(&b->mu);
    if (--b->active == 0) pthread_cond_signal(&b->cv_done);
    pthread_mutex_unlock(&b->mu);
  }
}

/* hand the workers a window to decode (or NULL) and a byte range to copy (or none) */
void bulk_start_b(struct bulk* b, bulk_window* w, const unsigned char* src, unsigned char* dst, size_t nbytes) {
  bulk_wait_b(b);                                 /* one job at a time */
  pthread_mutex_lock(&b->mu);
  b->work = w; b->next = 0;
  b->copy_src = src; b->copy_dst = dst; b->copy_bytes = nbytes; b->copy_next = 0; b->copy_row = b->next_copy_row;
  b->active = b->nth; b->gen++;
  pthread_cond_broadcast(&b->cv_work);
  pthread_mutex_unlock(&b->mu);
  b->in_b = w;
  b->busy = 1;
}
void bulk_wait_b(struct bulk* b) {
  if (!b->busy) return;
  const double t0 = now_s();
  pthread_mutex_lock(&b->mu);
  while (b->active) pthread_cond_wait(&b->cv_done, &b->mu);
  pthread_mutex_unlock(&b->mu);
  b->t_poolwait += now_s() - t0;
  b->busy = 0;
}

/* PCM of a finished slot -> caller memory.  An all-stereo window is one contiguous block: with `job` it is
 * left to the workers (returned through src/dst/nbytes), otherwise copied here. */
int bulk_collect(struct bulk* b, int slot, const unsigned char** jsrc, unsigned char** jdst, size_t* jbytes) {
  bulk_flight* f = &b->flight[slot];
  if (jbytes) *jbytes = 0;
  if (!f->active) return PDMP3_OK;
  const double t0 = now_s();
  const int wrc = pdmp3_hip_stream_wait(b->hs, slot);
  b->t_gpuwait += now_s() - t0;
  if (wrc != PDMP3_HIP_OK) return PDMP3_ERR;
  f->active = 0;
  if (f->direct) return PDMP3_OK;
  const unsigned char* src = (const unsigned char*)pdmp3_hip_stream_slot_pcm(b->hs, slot);
  const int to_device = b->pcm_pinned == 2;       /* memory the host cannot write: copies go through the engine */
  if (f->lsf) {
    /* stereo frames lie back to back (2304 bytes each), mono frames in pairs in the first half of a 4608-byte place */
    size_t off = 0;
    const size_t fb = f->all_stereo == 2 ? 2304 : 1152;
    for (int i = 0; i < f->n && off < f->dst_cap;) {
      const int run = f->all_stereo == 2 ? f->n - i : ((i & 1) || i + 1 == f->n ? 1 : 2);
      const unsigned char* from = f->all_stereo == 2 ? src + (size_t)i * 2304 : src + (size_t)(i >> 1) * 4608 + (size_t)(i & 1) * 1152;
      size_t n = (size_t)run * fb;
      if (n > f->dst_cap - off) n = f->dst_cap - off;
      if (to_device) { if (pdmp3_hip_copy_to_dest(f->dst + off, from, n) != PDMP3_HIP_OK) return PDMP3_ERR; }
      else memcpy(f->dst + off, from, n);
      off += (size_t)run * fb;
      i += run;
    }
    return PDMP3_OK;
  }
  if (f->all_stereo == 2 || (f->all_stereo == 1 && jbytes && !to_device)) {
    const size_t row = f->all_stereo == 2 ? 4608 : 2304;
    size_t n = (size_t)f->n * row;
    if (n > f->dst_cap) n = f->dst_cap;
    if (!n) return PDMP3_OK;
    if (to_device) return pdmp3_hip_copy_to_dest(f->dst, src, n) == PDMP3_HIP_OK ? PDMP3_OK : PDMP3_ERR;
    if (jbytes) { *jsrc = src; *jdst = f->dst; *jbytes = n; b->next_copy_row = row; }
    else memcpy(f->dst, src, n);
    return PDMP3_OK;
  }
  /* frames of both kinds (or mono frames without the pool): frame by frame, runs of equal frames as one copy */
  size_t off = 0;
  for (int i = 0; i < f->n;) {
    const size_t fb = 2304u * f->nch[i];
    int k = i + 1;
    if (fb == 4608) while (k < f->n && f->nch[k] == 2) k++;       /* stereo frames are dense in the slot */
    const size_t run = (size_t)(k - i) * fb;
    if (off < f->dst_cap) {
      const size_t n = run < f->dst_cap - off ? run : f->dst_cap - off;
      if (to_device) { if (pdmp3_hip_copy_to_dest(f->dst + off, src + (size_t)i * 4608, n) != PDMP3_HIP_OK) return PDMP3_ERR; }
      else memcpy(f->dst + off, src + (size_t)i * 4608, n);
    }
    off += run;
    i = k;
  }
  return PDMP3_OK;
}

/* a window whose frames all have the same channel count and that fits its destination goes there directly when
 * the destination is pinned */
static void flight_plan(struct bulk* b, bulk_flight* f) {
  const size_t row = (f->all_stereo == 2 ? 4608 : 2304) >> (f->lsf ? 1 : 0);
  f->direct = b->pcm_pinned && f->all_stereo != 0 && f->dst && (size_t)f->n * row <= f->dst_cap;
  /* (a window that cannot go there directly -- mixed mono / stereo frames, or the tail that does not fit -- is staged in
   * the slot's pinned buffer and copied by bulk_collect, through the engine when the destination is device memory) */
}

/* stage C + D of the window the workers have just finished */
int bulk_finish_b(struct bulk* b) {
  bulk_window* w = b->in_b;
  bulk_wait_b(b);
  if (!w) return PDMP3_OK;
  b->in_b = NULL;
  pdmp3_handle* id = b->id;
  bulk_flight* f = b->hs ? &b->flight[w->slot] : NULL;
  if (f) {
    f->dst = b->pcm_emitted < b->pcm_cap ? b->pcm + b->pcm_emitted : NULL;
    f->dst_cap = b->pcm_emitted < b->pcm_cap ? b->pcm_cap - b->pcm_emitted : 0;
    f->n = w->n; f->all_stereo = -1;
  }
  for (int i = 0; i < w->n; i++) {
    const frame_job* j = &w->jobs[i];
    apply_main(id, &j->hdr, &w->outs[i]);
    emit_records(id, &j->hdr, &j->si, j->reset, w->spectra + (size_t)i * 2304, w->side + (size_t)i * 4);
    const unsigned nch = j->hdr.mode == 3 ? 1 : 2;
    if (f) { f->nch[i] = (uint8_t)nch; f->all_stereo = f->all_stereo < 0 ? (int)nch : (f->all_stereo == (int)nch ? f->all_stereo : 0); }
    b->pcm_emitted += 2u * frame_samples(&j->hdr) * nch;
  }
  if (f) {
    f->lsf = w->n && w->jobs[0].hdr.ver != 0;     /* (bulk_push closes a window where the version -- or an LSF stream's channel count -- changes) */
    flight_plan(b, f);
    (void)pdmp3_hip_stream_set_lsf(b->hs, f->lsf);
    if (pdmp3_hip_stream_submit_to(b->hs, w->slot, w->n, f->direct ? f->dst : NULL, (f->all_stereo == 1 ? 2304 : 4608) >> (f->lsf ? 1 : 0)) != PDMP3_HIP_OK) {
      fprintf(stderr, "pdmp3: engine failure: %s\n", pdmp3_hip_last_error());
      return PDMP3_ERR;
    }
    f->active = 1;
  }
  return PDMP3_OK;
}

/* The window stage A has filled (k) goes to the workers, together with the PCM copy of window k-3, whose slot it
 * takes over; before that, window k-1 leaves the workers for stage C and the GPU. */
int bulk_rotate(struct bulk* b) {
  bulk_window* w = &b->win[b->cur];
  if (bulk_finish_b(b) != PDMP3_OK) return PDMP3_ERR;
  if (!w->n) return PDMP3_OK;
  const unsigned char* src = NULL;
  unsigned char* dst = NULL;
  size_t nbytes = 0;
  if (b->hs) {
    w->slot = (int)(b->windows % BULK_SLOTS);
    if (bulk_collect(b, w->slot, &src, &dst, &nbytes) != PDMP3_OK) return PDMP3_ERR;   /* window k-3 */
    w->spectra = pdmp3_hip_stream_slot_spectra(b->hs, w->slot);
    w->side = pdmp3_hip_stream_slot_side(b->hs, w->slot);
  } else {
    const size_t first = (size_t)b->frames - (size_t)w->n;
    if (first + (size_t)w->n > b->rec_cap) return PDMP3_ERR;
    w->spectra = b->rec_spectra + first * 2304;
    w->side = b->rec_side + first * 4;
  }
  bulk_start_b(b, w, src, dst, nbytes);
  b->windows++;
  b->cur ^= 1;
  b->win[b->cur].n = 0;
  return PDMP3_OK;
}

/* ---- bits mode: stage A writes side info + reservoir snapshot straight into the engine's pinned slot; scale-
 * factors, Huffman and the frame-to-frame merge run on the device (include/pdmp3_hip.h, submit_bits) ---- */
static void fill_frame_bits(const pdmp3_handle* id, pdmp3_frame_bits* fb, int newstream) {
  const frame_header* H = &id->hdr;
  const side_info* S = &id->si;
  const unsigned nch = H->mode == 3 ? 1 : 2;
  memset(fb, 0, sizeof *fb);
  fb->frame = (uint8_t)((H->sfreq & 3) | (H->mode << PDMP3_FR_MODE_SHIFT) | (H->mode_ext << PDMP3_FR_MODEEXT_SHIFT) |
                        (id->need_reset ? PDMP3_FR_RESET : 0) | (newstream ? PDMP3_FR_NEWSTREAM : 0));
  fb->iso = (uint8_t)id->iso;
  for (unsigned ch = 0; ch < nch; ch++)
    for (unsigned g4 = 0; g4 < 4; g4++) if (S->scfsi[ch][g4]) fb->scfsi[ch] |= (uint8_t)(1u << g4);
  for (unsigned gr = 0; gr < 2; gr++)
    for (unsigned ch = 0; ch < nch; ch++) {
      pdmp3_gc_bits* g = &fb->gc[gr * 2 + ch];
      g->part2_3_length = (uint16_t)S->part2_3_length[gr][ch];
      g->big_values = (uint16_t)S->big_values[gr][ch];
      g->global_gain = (uint8_t)S->global_gain[gr][ch];
      g->scalefac_compress = (uint8_t)S->scalefac_compress[gr][ch];
      g->flags = (uint8_t)((S->scalefac_scale[gr][ch] ? PDMP3_GC_SCALEFAC_SCALE : 0) |
                           (S->preflag[gr][ch] ? PDMP3_GC_PREFLAG : 0) |
                           (S->win_switch[gr][ch] ? PDMP3_GC_WIN_SWITCH : 0) |
                           ((S->block_type[gr][ch] & 3) << PDMP3_GC_BLOCK_TYPE_SHIFT) |
                           ((S->win_switch[gr][ch] && S->mixed[gr][ch]) ? PDMP3_GC_MIXED : 0));
      for (unsigned k = 0; k < 3; k++) {
        g->table_select[k] = (uint8_t)S->table_select[gr][ch][k];
        g->subblock_gain[k] = (uint8_t)S->subblock_gain[gr][ch][k];
      }
      g->region0_count = (uint8_t)S->region0_count[gr][ch];
      g->region1_count = (uint8_t)S->region1_count[gr][ch];
      g->count1table_select = (uint8_t)S->count1table_select[gr][ch];
    }
}

/* the main data a window's frames left in the caller's stream, into its pool (entries the scanner needed early have n = 0) */
/* (The pool is written once and read by the copy engine: non-temporal stores for the whole cache lines of an entry -- no
 *  read-for-ownership of 9 MB per window, nothing of it in the caches the scanners work in.  A frame's main data is about a
 *  kilobyte, its first and last partial line go the ordinary way.  PDMP3_BULK_GATHER_NT=0: memcpy.) */
static int g_gather_nt = -1;
void pool_gather(uint8_t* pool, const struct pool_copy* g, int n) {
#if defined(__x86_64__)
  int nt = __atomic_load_n(&g_gather_nt, __ATOMIC_RELAXED);
  if (nt < 0) { const char* e = getenv("PDMP3_BULK_GATHER_NT"); nt = !(e && *e == '0'); __atomic_store_n(&g_gather_nt, nt, __ATOMIC_RELAXED); }
  if (nt) {
    for (int i = 0; i < n; i++) {
      size_t len = g[i].n;
      if (!len) continue;
      unsigned char* dst = pool + g[i].dst;
      const unsigned char* src = g[i].src;
      if (len < 256) { memcpy(dst, src, len); continue; }
      const size_t head = (size_t)(-(uintptr_t)dst & 63);
      if (head) { memcpy(dst, src, head); dst += head; src += head; len -= head; }
      for (size_t lines = len >> 6; lines; lines--) {
        const __m128i a = _mm_loadu_si128((const __m128i*)src), b = _mm_loadu_si128((const __m128i*)src + 1);
        const __m128i c = _mm_loadu_si128((const __m128i*)src + 2), d = _mm_loadu_si128((const __m128i*)src + 3);
        _mm_stream_si128((__m128i*)dst, a); _mm_stream_si128((__m128i*)dst + 1, b);
        _mm_stream_si128((__m128i*)dst + 2, c); _mm_stream_si128((__m128i*)dst + 3, d);
        src += 64; dst += 64;
      }
      if (len & 63) memcpy(dst, src, len & 63);
    }
    _mm_sfence();
    return;
  }
#endif
  for (int i = 0; i < n; i++) if (g[i].n) memcpy(pool + g[i].dst, g[i].src, g[i].n);
}
/* (gh_mu held) one task off the queue and done; 0: the queue is empty */
static int gather_take_locked(struct bulk* b) {
  if (b->gq_tail == b->gq_head) return 0;
  const unsigned k = b->gq_tail++ % GATHER_QUEUE;
  uint8_t* pool = b->gq[k].pool;
  const struct pool_copy* list = (const struct pool_copy*)b->gq[k].list;
  const int n = b->gq[k].n, slot = b->gq[k].slot;
  pthread_mutex_unlock(&b->gh_mu);
  pool_gather(pool, list, n);
  pthread_mutex_lock(&b->gh_mu);
  if (--b->g_out[slot] == 0) pthread_cond_broadcast(&b->gh_done_cv);
  return 1;
}
void* gather_helper(void* arg) {
  struct bulk* b = (struct bulk*)arg;
  pthread_mutex_lock(&b->gh_mu);
  for (;;) {
    while (!b->gh_quit && b->gq_tail == b->gq_head) pthread_cond_wait(&b->gh_cv, &b->gh_mu);
    if (b->gh_quit) break;
    (void)gather_take_locked(b);
  }
  pthread_mutex_unlock(&b->gh_mu);
  return NULL;
}
/* the copies g[0, n) of `slot`'s window as tasks (a full queue: the caller does the copy itself) */
void gather_push(struct bulk* b, int slot, uint8_t* pool, const struct pool_copy* g, int n) {
  for (int lo = 0; lo < n; lo += GATHER_TASK_ENTRIES) {
    const int k = n - lo < GATHER_TASK_ENTRIES ? n - lo : GATHER_TASK_ENTRIES;
    pthread_mutex_lock(&b->gh_mu);
    if (b->gq_head - b->gq_tail >= GATHER_QUEUE) { pthread_mutex_unlock(&b->gh_mu); pool_gather(pool, g + lo, k); continue; }
    const unsigned q = b->gq_head++ % GATHER_QUEUE;
    b->gq[q].pool = pool; b->gq[q].list = g + lo; b->gq[q].n = k; b->gq[q].slot = slot;
    b->g_out[slot]++;
    pthread_cond_signal(&b->gh_cv);
    pthread_mutex_unlock(&b->gh_mu);
  }
}
/* until the slot's copies are complete; takes tasks (any slot's) while it waits */
static void gather_wait(struct bulk* b, int slot) {
  pthread_mutex_lock(&b->gh_mu);
  while (b->g_out[slot]) if (!gather_take_locked(b)) pthread_cond_wait(&b->gh_done_cv, &b->gh_mu);
  pthread_mutex_unlock(&b->gh_mu);
}
void* bulk_submitter(void* arg) {
  struct bulk* b = (struct bulk*)arg;
  for (;;) {
    pthread_mutex_lock(&b->sub_mu);
    while (b->sub_tail == b->sub_head && !b->sub_quit) pthread_cond_wait(&b->sub_cv, &b->sub_mu);
    if (b->sub_tail == b->sub_head) { pthread_mutex_unlock(&b->sub_mu); return NULL; }
    const int slot = b->sub_slot[b->sub_tail & 7], n = b->sub_n[b->sub_tail & 7], row = b->sub_row[b->sub_tail & 7];
    void* dst = b->sub_dst[b->sub_tail & 7];
    const size_t pool = b->sub_pool[b->sub_tail & 7];
    const int gn = b->sub_gath[b->sub_tail & 7];
    pthread_mutex_unlock(&b->sub_mu);
    const double t0 = now_s();
    if (pool) {
      if (gn > 0) gather_push(b, slot, pdmp3_hip_stream_slot_pool(b->hs, slot), b->gath[slot], gn);   /* (gn < 0: handed out while the window was filled) */
      gather_wait(b, slot);
    }
    const double t1 = now_s();
    pthread_mutex_lock(&b->sub_mu);                /* (whoever only waits for the stream's bytes to be free need not sit through the engine call) */
    b->sub_copied = b->sub_tail + 1;
    pthread_cond_broadcast(&b->sub_done_cv);
    pthread_mutex_unlock(&b->sub_mu);
    const int rc = pool ? pdmp3_hip_stream_submit_pool_to(b->hs, slot, n, pool, dst, row)
                        : pdmp3_hip_stream_submit_bits_to(b->hs, slot, n, dst, row);
    if (rc != PDMP3_HIP_OK) fprintf(stderr, "pdmp3: engine failure: %s\n", pdmp3_hip_last_error());
    const double t2 = now_s();
    b->t_sub_gather += t1 - t0; b->t_sub_call += t2 - t1;
    if (b->trace2) fprintf(stderr, "  submitter: %d frames to slot %d: taken at %.2f ms, copies %.0f us, engine call %.0f us\n", n, slot, (t0 - b->tr_t0) * 1e3, (t1 - t0) * 1e6, (t2 - t1) * 1e6);
    pthread_mutex_lock(&b->sub_mu);
    if (rc != PDMP3_HIP_OK) b->sub_rc = rc;
    b->sub_tail++;
    pthread_cond_broadcast(&b->sub_done_cv);
    pthread_mutex_unlock(&b->sub_mu);
  }
}
static long long sub_enqueue(struct bulk* b, int slot, int n, void* dst, int row, size_t pool_bytes, int gath_n) {
  pthread_mutex_lock(&b->sub_mu);
  const long long seq = b->sub_head;
  b->sub_slot[b->sub_head & 7] = slot; b->sub_n[b->sub_head & 7] = n;
  b->sub_dst[b->sub_head & 7] = dst; b->sub_row[b->sub_head & 7] = row; b->sub_pool[b->sub_head & 7] = pool_bytes;
  b->sub_gath[b->sub_head & 7] = gath_n;
  b->sub_head++;
  pthread_cond_signal(&b->sub_cv);
  pthread_mutex_unlock(&b->sub_mu);
  return seq;
}
static int sub_wait_seq(struct bulk* b, long long seq) {   /* window number `seq` of the queue has been handed to the GPU */
  if (!b->sub_started) return PDMP3_OK;
  pthread_mutex_lock(&b->sub_mu);
  while (b->sub_tail <= seq) pthread_cond_wait(&b->sub_done_cv, &b->sub_mu);
  const int rc = b->sub_rc;
  pthread_mutex_unlock(&b->sub_mu);
  return rc == PDMP3_HIP_OK ? PDMP3_OK : PDMP3_ERR;
}
/* every enqueued window's main data has been copied out of the caller's stream (the windows themselves may still be on
 * their way to the GPU: a failure there shows at the next call or at the wait) */
int sub_drain_copied(struct bulk* b) {
  if (!b->sub_started) return PDMP3_OK;
  pthread_mutex_lock(&b->sub_mu);
  while (b->sub_copied < b->sub_head) pthread_cond_wait(&b->sub_done_cv, &b->sub_mu);
  const int rc = b->sub_rc;
  pthread_mutex_unlock(&b->sub_mu);
  return rc == PDMP3_HIP_OK ? PDMP3_OK : PDMP3_ERR;
}
int sub_drain(struct bulk* b) {            /* every enqueued window has been handed to the GPU */
  if (!b->sub_started) return PDMP3_OK;
  pthread_mutex_lock(&b->sub_mu);
  while (b->sub_tail != b->sub_head) pthread_cond_wait(&b->sub_done_cv, &b->sub_mu);
  const int rc = b->sub_rc;
  pthread_mutex_unlock(&b->sub_mu);
  return rc == PDMP3_HIP_OK ? PDMP3_OK : PDMP3_ERR;
}

/* make the slot of window `windows` writable: its previous occupant (window - BULK_SLOTS) must be off the GPU; its PCM
 * goes home on the worker pool while stage A fills the slot's input side */
/* A stream's first windows are short -- cap / 8, cap / 8, cap / 4, cap / 2, then cap frames each: the GPU has something to
 * do after an eighth of a window's scan instead of a whole one (0.33 ms of a 7 ms decode at 4096 frames, twice that at
 * 8192), and the four together are exactly one full window, so every later window starts where it would have.  Device
 * Huffman with the compact upload only, and only for windows large enough to notice. */
static int win_ramp(const struct bulk* b) { return b->ramp_on && b->bits_mode && b->pool_mode && !b->win_sink && b->target >= 1024 && b->target % 8 == 0; }
static int win_frames(const struct bulk* b, long long w) {
  if (b->win_sink) return b->cap;                 /* (a split scan's private window) */
  if (!win_ramp(b) || w >= 4) return b->cur_target > 0 ? b->cur_target : b->target;
  return w < 2 ? b->target / 8 : w == 2 ? b->target / 4 : b->target / 2;
}
/* bytes the scanner itself puts into the window's pool (a segment's image of the reservoir buffer, a frame's own image):
 * straight into the pool, and for a split scan's private window also into its arena and its copy list */
static int pool_literal(struct bulk* b, size_t dst, const uint8_t* src, size_t n) {
  memcpy(b->res_dst + dst, src, n);
  if (!b->win_sink) return PDMP3_OK;
  pre_window* w = b->pw_cur;
  if (w->arena_len + n > PW_ARENA_BYTES || b->gath_n >= b->gath_cap - 1) return PDMP3_ERR;
  memcpy(w->arena + w->arena_len, src, n);
  struct pool_copy* g = &b->gath_cur[b->gath_n++];
  g->src = w->arena + w->arena_len; g->dst = (uint32_t)dst; g->n = (uint32_t)n;
  w->arena_len += n;
  return PDMP3_OK;
}
int bits_open_window(struct bulk* b) {
  b->bits_n = 0;
  b->bits_open = 1;
  b->pool_tail = 0; b->need_segment = 1; b->seg_first = 0; b->cur_explicit = 0; b->cur_staged = 0; b->sky_n = 0;
  b->gath_n = 0;
  if (b->win_sink) {                              /* split scan: a private window */
    b->pw_cur = pw_new_in(b->pc, b->cap, b->stream_win);
    if (!b->pw_cur) return PDMP3_ERR;
    b->bits_dst = b->pw_cur->bits; b->desc_dst = b->pw_cur->desc;
    b->gath_cur = (struct pool_copy*)b->pw_cur->gath; b->gath_cap = b->cap + BULK_GATH_EXTRA;
    b->res_dst = b->priv_pool;
    b->pool_cap = (size_t)b->cap * RESERVOIR_BYTES + PDMP3_POOL_SLACK_BYTES;
    return PDMP3_OK;
  }
  if (!b->hs) {                                   /* parse only: one "window" = the caller's arrays */
    b->bits_dst = b->rec_bits;
    b->res_dst = b->rec_res;
    b->desc_dst = b->rec_desc; b->pool_cap = b->rec_pool_cap;
    if (b->pool_mode) {
      free(b->gath[0]);
      b->gath_cur = b->gath[0] = (struct pool_copy*)malloc((b->rec_cap + 1) * sizeof(struct pool_copy));
      if (!b->gath_cur) return PDMP3_ERR;
      b->gath_cap = b->rec_cap + 1;
    }
    return PDMP3_OK;
  }
  b->bits_slot = (int)(b->windows % BULK_SLOTS);
  const unsigned char* src; unsigned char* dst; size_t nbytes;
  if (b->flight[b->bits_slot].active) {           /* (long done: BULK_SLOTS windows ago) */
    const double t0 = now_s();
    const int rc = sub_wait_seq(b, b->flight[b->bits_slot].sub_seq);   /* (not the windows queued after it) */
    b->t_subwait += now_s() - t0;
    if (rc != PDMP3_OK) return PDMP3_ERR;
  }
  if (bulk_collect(b, b->bits_slot, &src, &dst, &nbytes) != PDMP3_OK) return PDMP3_ERR;
  if (nbytes) bulk_start_b(b, NULL, src, dst, nbytes);
  b->g_pushed[b->bits_slot] = 0;
  for (int i = 0; i < b->slot_arena_n[b->bits_slot]; i++) free(b->slot_arena[b->bits_slot][i]);       /* (its window was gathered long ago) */
  b->slot_arena_n[b->bits_slot] = 0;
  b->bits_dst = pdmp3_hip_stream_slot_bits(b->hs, b->bits_slot);
  b->res_dst = pdmp3_hip_stream_slot_reservoir(b->hs, b->bits_slot);
  if (b->pool_mode) {
    b->desc_dst = pdmp3_hip_stream_slot_rowdesc(b->hs, b->bits_slot);
    b->pool_cap = pdmp3_hip_stream_pool_bytes(b->hs);
    b->gath_cur = b->gath[b->bits_slot]; b->gath_cap = b->cap + BULK_GATH_EXTRA * (PAR_MAX_BATCH + 1);
    if (!b->desc_dst || !b->gath_cur) return PDMP3_ERR;
  }
  return b->bits_dst && b->res_dst ? PDMP3_OK : PDMP3_ERR;
}

/* ---- compact bits input: the window's pool (include/pdmp3_hip.h, pdmp3_row_desc) ----
 * While a segment runs, id->main_vec is NOT updated: it keeps the buffer as it was when the segment began (that image
 * is in the pool at seg_s_off), and "the buffer's valid bytes [0, main_top) are the last main_top bytes of the pool"
 * holds.  pool_materialize() brings main_vec up to date again from the segment's frames (the same rule the device
 * applies, unpack_core.h row_byte): before anything irregular touches the buffer, and when the window closes. */
/* pool bytes [lo, hi) are needed now: the copies that are still only noted and touch them (the list is in pool order) */
static void pool_ensure(struct bulk* b, size_t lo, size_t hi) {
  for (int i = b->gath_n - 1; i >= 0; i--) {
    struct pool_copy* g = &b->gath_cur[i];
    if ((size_t)g->dst + g->n <= lo && g->n) break;           /* (entries done earlier have n = 0: keep looking) */
    if (g->n && g->dst < hi) {
      memcpy(b->res_dst + g->dst, g->src, g->n);
      if (!b->win_sink) g->n = 0;                 /* (a split scan's private pool is not the one that goes up: the entry stays) */
      else if ((size_t)g->dst < lo) break;
    }
  }
}
void pool_materialize(struct bulk* b) {
  pdmp3_handle* id = b->id;
  if (b->need_segment) return;                    /* main_vec is live */
  unsigned covered = 0;
  if (b->sky_n)                                   /* the last frame, then up its links: each hop has a larger top */
    for (const pdmp3_row_desc* d = &b->desc_dst[b->sky[b->sky_n - 1]];; d -= d->up) {
      pool_ensure(b, (size_t)d->row_off + covered, (size_t)d->row_off + d->top);
      memcpy(id->main_vec + covered, b->res_dst + d->row_off + covered, d->top - covered);
      covered = d->top;
      if (!d->up) break;
    }
  b->need_segment = 1;
}


/* Get_Main_Data (P:1096-1122) of the frame being staged, into the pool.  Same return codes and the same effect on
 * main_top and the ring as fill_reservoir. */
int fill_reservoir_pool(pdmp3_handle* id, unsigned size, unsigned begin) {
  struct bulk* b = id->pool_sink;
  if (!b->bits_open && bits_open_window(b) != PDMP3_OK) { b->failed = 1; return PDMP3_ERR; }
  if (b->cur_staged) { b->failed = 1; return PDMP3_ERR; }   /* (every frame staged here is pushed: the pool is the only copy) */
  const int regular = begin <= id->main_top && size <= ring_filled(id) && begin + size <= sizeof id->main_vec;
  if (!regular) {
    /* reservoir underflow (H9), a frame the ring does not hold completely (H18), or more bytes than the buffer
     * takes: the reference's buffer arithmetic on the real buffer; a frame that is decoded all the same carries its
     * own image of the result */
    pool_materialize(b);
    b->cur_explicit = 1;
    return fill_reservoir(id, size, begin);
  }
  if (b->need_segment) {                          /* the buffer as it is now, then its valid tail once more */
    const unsigned h = id->main_top < 511 ? id->main_top : 511;
    b->seg_s_off = (uint32_t)b->pool_tail;
    if (pool_literal(b, b->pool_tail, id->main_vec, RESERVOIR_BYTES) != PDMP3_OK) { b->failed = 1; return PDMP3_ERR; }
    b->pool_tail += RESERVOIR_BYTES;
    if (h && pool_literal(b, b->pool_tail, id->main_vec + id->main_top - h, h) != PDMP3_OK) { b->failed = 1; return PDMP3_ERR; }
    b->pool_tail += h;
    b->seg_first = b->bits_n;
    b->sky_n = 0;
    b->need_segment = 0;
  }
  b->cur_row_off = (uint32_t)(b->pool_tail - begin);
  b->cur_top = begin + size;
  b->cur_explicit = 0;
  b->cur_staged = 1;
  if (id->vsrc) {                                 /* the bytes stay where they are for now (pool_gather) */
    /* the copy list holds one entry per frame and BULK_GATH_EXTRA literals (a split scan's private window; an engine
     * window that private windows are stitched into: that many per private window): never write past it, whatever
     * PW_ARENA_BYTES / RESERVOIR_BYTES let through (the window fails; the stream then takes the one-thread scan) */
    if (b->gath_n >= b->gath_cap) { b->failed = 1; return PDMP3_ERR; }
    struct pool_copy* g = &b->gath_cur[b->gath_n++];
    g->src = id->vsrc + id->vfed - ring_filled(id); g->dst = (uint32_t)b->pool_tail; g->n = size;
    id->istart = (id->istart + size) % INBUF_SIZE;
    id->processed += size;
  } else ring_take(id, b->res_dst + b->pool_tail, size);
  b->pool_tail += size;
  id->main_top = begin + size;
  return PDMP3_OK;
}

int bits_close_window(struct bulk* b) {
  if (!b->bits_open) return PDMP3_OK;
  if (b->pool_mode) pool_materialize(b);          /* the next window starts from the live buffer */
  b->bits_open = 0;
  if (!b->bits_n) return PDMP3_OK;
  if (b->hs) {
    bulk_wait_b(b);                               /* the slot's old PCM has been copied out */
    bulk_flight* f = &b->flight[b->bits_slot];
    f->n = b->bits_n;
    f->dst = b->pcm_emitted < b->pcm_cap ? b->pcm + b->pcm_emitted : NULL;
    f->dst_cap = b->pcm_emitted < b->pcm_cap ? b->pcm_cap - b->pcm_emitted : 0;
    f->all_stereo = f->nch[0];
    for (int i = 0; i < f->n; i++) {
      if (f->nch[i] != f->nch[0]) f->all_stereo = 0;
      b->pcm_emitted += 2304u * f->nch[i];
    }
    const double t0 = now_s();
    flight_plan(b, f);
    f->sub_seq = sub_enqueue(b, b->bits_slot, b->bits_n, f->direct ? f->dst : NULL, f->all_stereo == 1 ? 2304 : 4608, b->pool_mode ? b->pool_tail : 0, b->g_pushed[b->bits_slot] ? -1 : b->gath_n);
    b->t_submit += now_s() - t0;
    f->active = 1;
  }
  b->windows++;
  b->stream_win++;
  return PDMP3_OK;
}

static int bits_push(struct bulk* b) {
  pdmp3_handle* id = b->id;
  if (!b->bits_open && bits_open_window(b) != PDMP3_OK) { b->failed = 1; return PDMP3_ERR; }
  if (!b->hs && !b->win_sink && (size_t)b->frames > b->rec_cap) { b->failed = 1; return PDMP3_ERR; }
  const int i = b->bits_n++;
  if (id->fb_valid) {                             /* read_side_info_bits has built the record */
    const frame_header* H = &id->hdr;
    id->fb_cur.frame = (uint8_t)((H->sfreq & 3) | (H->mode << PDMP3_FR_MODE_SHIFT) | (H->mode_ext << PDMP3_FR_MODEEXT_SHIFT) |
                                 (id->need_reset ? PDMP3_FR_RESET : 0) | (b->frames == 1 && !b->carry ? PDMP3_FR_NEWSTREAM : 0));
    b->bits_dst[i] = id->fb_cur;
  } else
  fill_frame_bits(id, &b->bits_dst[i], b->frames == 1 && !b->carry);   /* a fresh handle's parse state is zero */
  id->need_reset = 0;
  if (b->pool_mode) {
    pdmp3_row_desc* d = &b->desc_dst[i];
    b->cur_staged = 0;
    if (b->cur_explicit) {                        /* its own image of the buffer (main_vec is live here) */
      d->row_off = d->s_off = (uint32_t)b->pool_tail;
      d->top = RESERVOIR_BYTES; d->back = 0; d->up = 0; d->reserved = 0;
      if (pool_literal(b, b->pool_tail, id->main_vec, RESERVOIR_BYTES) != PDMP3_OK) { b->failed = 1; return PDMP3_ERR; }
      b->pool_tail += RESERVOIR_BYTES;
      b->cur_explicit = 0;
    } else {
      d->row_off = b->cur_row_off; d->s_off = b->seg_s_off;
      d->top = (uint16_t)b->cur_top; d->back = (uint16_t)(i - b->seg_first); d->reserved = 0;
      /* previous frame of the segment with a larger top: the skyline seen from this frame (a stack of strictly
       * decreasing tops, so never deeper than 2064) */
      while (b->sky_n && b->desc_dst[b->sky[b->sky_n - 1]].top <= d->top) b->sky_n--;
      d->up = (uint16_t)(b->sky_n ? i - b->sky[b->sky_n - 1] : 0);
      b->sky[b->sky_n++] = i;
      if (i - b->seg_first >= 65000) pool_materialize(b);   /* (`back` / `up` are 16 bits: a very long window starts a new segment) */
    }
  } else memcpy(b->res_dst + (size_t)i * RESERVOIR_BYTES, id->main_vec, RESERVOIR_BYTES);
  if (b->hs) b->flight[b->bits_slot].nch[i] = (uint8_t)(id->hdr.mode == 3 ? 1 : 2);
  if (b->win_sink) b->pw_cur->nch[i] = (uint8_t)(id->hdr.mode == 3 ? 1 : 2);
  const int full = b->bits_n >= win_frames(b, b->stream_win) || (b->pool_mode && b->pool_tail + POOL_ROOM > b->pool_cap);
  if (b->win_sink && full && pw_close_window(b) != PDMP3_OK) { b->failed = 1; return PDMP3_ERR; }
  if (b->win_sink) return PDMP3_OK;
  if (b->hs && full && bits_close_window(b) != PDMP3_OK) { b->failed = 1; return PDMP3_ERR; }
  if (!b->hs && !b->win_sink && b->pool_mode && b->pool_tail + POOL_ROOM > b->pool_cap) { b->failed = 1; return PDMP3_ERR; }
  return PDMP3_OK;
}

static int bulk_push(struct bulk* b) {
  pdmp3_handle* id = b->id;
  b->frames++;
  if (b->count_only) { id->need_reset = 0; return PDMP3_OK; }
  if (b->bits_mode) return bits_push(b);
  bulk_window* w = &b->win[b->cur];
  /* the engine takes LSF frames in launches of their own, all of one channel count: such a frame opens a new window */
  if (w->n && (id->hdr.ver != w->jobs[0].hdr.ver || (id->hdr.ver && (id->hdr.mode == 3) != (w->jobs[0].hdr.mode == 3)))) {
    if (bulk_rotate(b) != PDMP3_OK) { b->failed = 1; return PDMP3_ERR; }
    w = &b->win[b->cur];
  }
  frame_job* j = &w->jobs[w->n++];
  j->hdr = id->hdr;
  j->si = id->si;
  j->reset = (uint8_t)id->need_reset;
  id->need_reset = 0;
  memcpy(j->res, id->main_vec, RESERVOIR_BYTES);
  if (w->n == b->cap && bulk_rotate(b) != PDMP3_OK) { b->failed = 1; return PDMP3_ERR; }
  return PDMP3_OK;
}

/* stage A: the CLI's loop (P:2566-2583) over a memory buffer.  Returns the PCM bytes pdmp3() would write. */
long long bulk_drive(struct bulk* b, const unsigned char* mp3, size_t n) {
  pdmp3_handle* id = b->id;
  pdmp3_open_feed(id);
  id->vsrc = mp3; id->vfed = 0;                   /* the ring's bytes are the buffer's: only its indices move */
  size_t fed = 0, done, total = 0;
  int res;
  while ((res = read_impl_sink(id, INBUF_SIZE, &done, b)) != PDMP3_ERR) {
    total += done;
    /* More bytes consumed than were ever fed: the ring is being replayed.  pdmp3_feed parks iend AT INBUF_SIZE
     * when a feed ends exactly at the end of the ring (P:2410-2417); a frame that then ends exactly there (only
     * 1152-byte frames can: 32 kHz / 256 kbps, the H10 limit) wraps the read index to 0 != iend and the ring
     * looks full of its own stale contents (P:1464-1474).  The reference -- and pdmp3_read / pdmp3(), which keep
     * its behaviour -- then emit the last 16 KiB again, often forever.  There is no finite reference output to
     * match, so the whole-stream entry points stop here. */
    if (id->processed > fed) { id->vsrc = NULL; b->failed = 2; return PDMP3_BULK_REPLAY; }
    if (res == PDMP3_NEED_MORE) {
      size_t take = n - fed < 4096 ? n - fed : 4096;
      if (!take) break;
      if (id->vsrc && take > ring_free_logical(id)) {
        /* the CLI drops a feed the ring has no room for (H16, after an underflow's NEED_MORE): from here on the ring's
         * bytes are no longer the buffer's at `processed` -- give the ring its real contents and go on with copies */
        for (unsigned k = 0, f = ring_filled(id); k < f; k++) id->in[(id->istart + k) % INBUF_SIZE] = id->vsrc[id->vfed - f + k];
        id->vsrc = NULL;
      }
      (void)pdmp3_feed(id, mp3 + fed, take);
      fed += take;
    }
  }
  id->vsrc = NULL;
  return (long long)total;
}

