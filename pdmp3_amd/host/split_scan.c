/* split_scan.c -- libpdmp3.so: the whole-stream decoder's scan split over threads (round 4): hop threads, pre-pass,
 * scanners into private windows, the stitcher (par_drive); and the host tests' hook that dumps the windows.
 * See host_internal.h for the map of the library. */
#include "bulk_internal.h"

/* ------------------------------------------------------------------------ */
/* Split scan (round 4).  Stage A is ~300 cycles of bit-field parsing and      */
/* index arithmetic per frame on ONE thread (profiles/r04_scan_sections.txt),  */
/* and with the PCM left on the device it is what bounds the whole-stream      */
/* decoder.  What is strictly sequential in it is little: a frame's position   */
/* follows from the header before it, the ring's indices from the CLI's feed   */
/* cadence, the reservoir's fill from the frame before.  So: a PRE-PASS hops    */
/* from header to header with the real ring arithmetic (pdmp3_feed on a        */
/* scratch handle), checks that every frame takes the scanner's regular path    */
/* and leaves, at a few window boundaries, what a scanner needs to start there  */
/* -- ring indices, reservoir fill, which earlier frames' bytes are still in    */
/* the reservoir buffer, which frames last set the side-info fields the         */
/* reference leaves stale (H20); SCANNER threads run the unchanged stage-A code */
/* from those points into private windows (pre_window); the calling thread      */
/* moves the windows into the engine's slots in stream order.  Anything the     */
/* pre-pass does not recognise as regular (resync, underflow H9, a frame the    */
/* ring does not hold H18, a dropped feed H16, a replayed ring) makes the whole */
/* stream go the one-thread way from the start: results are the sequential      */
/* scanner's by construction, bit for bit (tests compare).                      */
/* ------------------------------------------------------------------------ */

static void pw_destroy(pre_window* w) {
  if (!w) return;
  free(w->bits); free(w->desc); free(w->nch); free(w->gath); free(w->arena);
  free(w);
}
pre_window* pw_new_in(par_cache* pc, int cap, long long index) {
  pre_window* w = NULL;
  if (pc) {
    pthread_mutex_lock(&pc->mu);
    while (pc->n_spare && !w) {
      w = pc->spare[--pc->n_spare];
      if (w->cap < cap) { pw_destroy(w); w = NULL; }    /* (too small for this stream's private windows; a larger one does) */
    }
    pthread_mutex_unlock(&pc->mu);
  }
  if (w) {
    w->index = index; w->n = w->gath_n = 0; w->pool_tail = 0; w->arena_len = 0;
    if (!w->arena && !(w->arena = (uint8_t*)malloc(PW_ARENA_BYTES))) { pw_destroy(w); return NULL; }    /* (it went with a slot) */
    return w;
  }
  w = (pre_window*)calloc(1, sizeof *w);
  if (!w) return NULL;
  w->index = index; w->cap = cap; w->home = pc;
  w->bits = (pdmp3_frame_bits*)malloc((size_t)cap * sizeof(pdmp3_frame_bits));
  w->desc = (pdmp3_row_desc*)malloc((size_t)cap * sizeof(pdmp3_row_desc));
  w->nch = (uint8_t*)malloc((size_t)cap);
  w->gath = malloc(((size_t)cap + BULK_GATH_EXTRA) * sizeof(struct pool_copy));
  w->arena = (uint8_t*)malloc(PW_ARENA_BYTES);
  if (!w->bits || !w->desc || !w->nch || !w->gath || !w->arena) { pw_destroy(w); return NULL; }
  return w;
}
void pw_free(pre_window* w) {
  if (!w) return;
  par_cache* pc = w->home;
  if (pc) {
    pthread_mutex_lock(&pc->mu);
    if (pc->n_spare < (int)(sizeof pc->spare / sizeof pc->spare[0])) { pc->spare[pc->n_spare++] = w; w = NULL; }
    pthread_mutex_unlock(&pc->mu);
  }
  pw_destroy(w);
}
static par_cache* pc_new(void) {
  par_cache* pc = (par_cache*)calloc(1, sizeof *pc);
  if (pc) {
    pthread_mutex_init(&pc->mu, NULL);
    pthread_mutex_init(&pc->crew_mu, NULL); pthread_cond_init(&pc->crew_cv, NULL); pthread_cond_init(&pc->crew_done_cv, NULL);
  }
  return pc;
}
#define CREW_MAX (PAR_MAX_SCANNERS + PAR_MAX_SEGS)
static void* crew_main(void* arg) {
  par_cache* pc = (par_cache*)arg;
  pthread_mutex_lock(&pc->crew_mu);
  for (;;) {
    while (!pc->crew_quit && pc->job_tail == pc->job_head) pthread_cond_wait(&pc->crew_cv, &pc->crew_mu);
    if (pc->job_tail == pc->job_head) break;            /* (quit, nothing left) */
    const unsigned k = pc->job_tail++ % CREW_MAX;
    void* (*fn)(void*) = pc->jobs[k].fn;
    void* a = pc->jobs[k].arg;
    int* left = pc->jobs[k].left;
    pthread_mutex_unlock(&pc->crew_mu);
    (void)fn(a);
    pthread_mutex_lock(&pc->crew_mu);
    pc->crew_busy--;
    (*left)--;                                          /* (under crew_mu: whoever waits for 0 may free what `left` is part of) */
    pthread_cond_broadcast(&pc->crew_done_cv);
  }
  pthread_mutex_unlock(&pc->crew_mu);
  return NULL;
}
/* fn(arg) on a thread of its own, at once (the jobs of a stream wait for each other: each needs a thread); *left counts
 * the stream's jobs that have not returned.  -1: no thread to be had. */
static int crew_run(par_cache* pc, void* (*fn)(void*), void* arg, int* left) {
  pthread_mutex_lock(&pc->crew_mu);
  if (pc->crew_busy >= pc->crew_n) {
    if (pc->crew_n >= CREW_MAX || pthread_create(&pc->crew[pc->crew_n], NULL, crew_main, pc) != 0) { pthread_mutex_unlock(&pc->crew_mu); return -1; }
    if (CPU_COUNT(&pc->near_gpu) > 0) (void)pthread_setaffinity_np(pc->crew[pc->crew_n], sizeof pc->near_gpu, &pc->near_gpu);
    pc->crew_n++;
  }
  const unsigned k = pc->job_head++ % CREW_MAX;
  pc->jobs[k].fn = fn; pc->jobs[k].arg = arg; pc->jobs[k].left = left;
  pc->crew_busy++;
  (*left)++;
  pthread_mutex_unlock(&pc->crew_mu);
  return 0;
}
/* the jobs queued so far may start: ONE wake-up for all of them (a signal per job is a system call per job on the calling
 * thread -- eleven of them in front of a file of a few minutes that is scanned in 0.2 ms) */
static void crew_kick(par_cache* pc) {
  pthread_mutex_lock(&pc->crew_mu);
  pthread_cond_broadcast(&pc->crew_cv);
  pthread_mutex_unlock(&pc->crew_mu);
}
static void crew_wait(par_cache* pc, int* left) {
  pthread_mutex_lock(&pc->crew_mu);
  while (*left) pthread_cond_wait(&pc->crew_done_cv, &pc->crew_mu);
  pthread_mutex_unlock(&pc->crew_mu);
}
void pc_free(par_cache* pc) {
  if (!pc) return;
  pthread_mutex_lock(&pc->crew_mu); pc->crew_quit = 1; pthread_cond_broadcast(&pc->crew_cv); pthread_mutex_unlock(&pc->crew_mu);
  for (int i = 0; i < pc->crew_n; i++) pthread_join(pc->crew[i], NULL);
  pthread_mutex_destroy(&pc->crew_mu); pthread_cond_destroy(&pc->crew_cv); pthread_cond_destroy(&pc->crew_done_cv);
  for (int i = 0; i < pc->n_spare; i++) pw_destroy(pc->spare[i]);
  for (int j = 0; j < PAR_MAX_SEGS; j++) free(pc->seg_rec[j]);
  for (int k = 0; k < PAR_MAX_SCANNERS; k++) { free(pc->scan[k].wb); free(pc->scan[k].id); free(pc->scan[k].pool); }
  free(pc->rec);
  pthread_mutex_destroy(&pc->mu);
  free(pc);
}
/* the private window is complete: the reservoir buffer is brought up to date for the next one (as bits_close_window
 * does) and the window goes to whoever stitches the stream together */
int pw_close_window(struct bulk* b) {
  if (!b->bits_open) return PDMP3_OK;
  pool_materialize(b);
  b->bits_open = 0;
  pre_window* w = b->pw_cur;
  b->pw_cur = NULL;
  if (!b->bits_n) { pw_free(w); return PDMP3_OK; }
  w->n = b->bits_n; w->gath_n = b->gath_n; w->pool_tail = b->pool_tail;
  w->t_take = b->tr_take; w->t_begin = b->tr_begin; w->t_done = now_s();
  struct par_scan* P = b->par;
  pthread_mutex_lock(&P->mu);
  if (w->index < P->win_cap && !P->win[w->index]) { P->win[w->index] = w; w = NULL; }
  pthread_cond_broadcast(&P->cv);
  pthread_mutex_unlock(&P->mu);
  if (w) { pw_free(w); return PDMP3_ERR; }          /* (cannot happen: more windows than the stream has bytes for) */
  b->windows++;
  b->stream_win++;
  return PDMP3_OK;
}

void header_fields(uint32_t h, frame_header* H) {
  H->id = (h >> 19) & 1; H->layer = 4 - ((h >> 17) & 3); H->protection = (h >> 16) & 1;
  H->bitrate_index = (h >> 12) & 15; H->sfreq = (h >> 10) & 3; H->padding = (h >> 9) & 1;
  H->mode = (h >> 6) & 3; H->mode_ext = (h >> 4) & 3;
  H->ver = 0;                                      /* (the split scan and the window estimates are MPEG-1's: bits mode never takes LSF) */
}
static inline uint32_t be32(const unsigned char* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

/* A frame's own fields at offset x (at least 40 bytes of stream from there); -1: not a frame the regular path takes */
static int hop_parse(const unsigned char* mp3, size_t x, hop1* r) {
  const uint32_t hw = be32(mp3 + x);
  if ((hw & 0xfff00000u) != 0xfff00000u) return -1;     /* the scanner would search for the next sync */
  frame_header H;
  header_fields(hw, &H);
  if (H.id != 1 || H.bitrate_index == 0 || H.bitrate_index == 15 || H.sfreq == 3 || H.layer != 3) return -1;
  const unsigned nch = H.mode == 3 ? 1 : 2, nbytes = nch == 1 ? 17 : 32, crc = H.protection == 0 ? 2 : 0;
  const unsigned fb = frame_bytes(&H);
  if (fb > 2000) return -1;
  const unsigned char* v = mp3 + x + 4 + crc;
  const unsigned begin = ((unsigned)v[0] << 1) | (v[1] >> 7);
  const unsigned size = fb - nbytes - 4 - crc;
  if (begin + size > RESERVOIR_BYTES) return -1;        /* (overflow of the reservoir buffer) */
  unsigned ws = 0;
  for (unsigned gr = 0; gr < 2; gr++)                   /* which frame last set the fields the reference leaves stale (H20) */
    for (unsigned ch = 0; ch < nch; ch++) {
      const unsigned pos = (nch == 1 ? 18u : 20u) + 59u * (gr * nch + ch) + 33u;
      ws |= ((v[pos >> 3] >> (7 - (pos & 7))) & 1u) << (gr * 2 + ch);
    }
  r->x = (uint32_t)x; r->fb = (uint16_t)fb; r->begin = (uint16_t)begin; r->top = (uint16_t)(begin + size);
  r->nch = (uint8_t)nch; r->crc = (uint8_t)crc; r->ws = (uint8_t)ws; r->pad = 0;
  return 0;
}
/* this is a chain of dependent cache misses -- where the next header is follows from this one -- unless the lines are
 * asked for ahead of time: frames of a constant-bitrate stream are as long as each other to within the padding byte, so
 * the header and side info of the 12th frame from here lie within 12 bytes of x + 12 fb (two lines cover them); on a
 * variable-bitrate stream the guess is wrong and costs nothing */
static inline void hop_prefetch(const unsigned char* mp3, size_t x, unsigned fb) {
  __builtin_prefetch(mp3 + x + 12u * fb, 0, 3);
  __builtin_prefetch(mp3 + x + 12u * fb + 64, 0, 3);
}
#define HOP_END_BYTES 1152u   /* the CLI's loop stops asking once fewer bytes than this are left (H10) */
static long long seg_wait_start(pre_seg* S) {           /* (found within microseconds of the thread's start) */
  long long v;
  while ((v = __atomic_load_n(&S->x_start, __ATOMIC_ACQUIRE)) == -2) {
    if (S->P->abort || S->P->quit) return -1;
    sched_yield();
  }
  return v;
}
static void* par_hop_thread(void* arg) {
  pre_seg* S = (pre_seg*)arg;
  struct par_scan* P = S->P;
  const unsigned char* mp3 = P->mp3;
  const size_t n = P->n;
  /* a header like the stream's first (MPEG-1 Layer III, same sampling rate) that three more follow */
  const uint32_t like = be32(mp3) & 0xfffe0c00u;
  long long found = -1;
  const size_t stop = S->guess + 65536 < n ? S->guess + 65536 : n;
  for (size_t x = S->guess; x < stop && x + 4u * 2000u + 64u <= n; x++) {
    if (mp3[x] != 0xff || (be32(mp3 + x) & 0xfffe0c00u) != like) continue;
    size_t y = x;
    int ok = 1;
    for (int k = 0; k < 4 && ok; k++) {
      hop1 r;
      if ((be32(mp3 + y) & 0xfffe0c00u) != like || hop_parse(mp3, y, &r) != 0) ok = 0;
      else y += r.fb;
    }
    if (ok) { found = (long long)x; break; }
  }
  __atomic_store_n(&S->x_start, found, __ATOMIC_RELEASE);
  S->t_sync = now_s() - P->t0;
  int state = SEG_BAD;
  if (found >= 0) {
    long long next_start = -3;                          /* (-3: the last part) */
    if (S->j + 1 < P->J) next_start = seg_wait_start(&P->seg[S->j + 1]);
    size_t x = (size_t)found;
    long long c = 0;
    for (;;) {
      if (next_start == -1 || P->abort) break;          /* (the next part has no start: given up) */
      if (next_start >= 0 && x >= (size_t)next_start) { state = x == (size_t)next_start ? SEG_AT_NEXT : SEG_BAD; break; }
      if (n - x < HOP_END_BYTES) { state = SEG_AT_END; break; }
      if (c >= S->cap || hop_parse(mp3, x, &S->rec[c]) != 0) break;
      hop_prefetch(mp3, x, S->rec[c].fb);
      x += S->rec[c].fb;
      c++;
      if (!(c & 7)) __atomic_store_n(&S->count, c, __ATOMIC_RELEASE);
    }
    __atomic_store_n(&S->count, c, __ATOMIC_RELEASE);
  }
  S->t_done = now_s() - P->t0;
  __atomic_store_n(&S->state, state, __ATOMIC_RELEASE);
  return NULL;
}

/* The pre-pass.  Returns 0 when the whole stream is regular (P->n_frames frames), -1 otherwise. */
static int par_prepass(struct par_scan* P) {
  const unsigned char* mp3 = P->mp3;
  const size_t n = P->n;
  pdmp3_handle* h = (pdmp3_handle*)calloc(1, sizeof *h);
  int* sky = (int*)malloc((RESERVOIR_BYTES + 2) * sizeof(int));
  if (!h || !sky) { free(h); free(sky); return -1; }
  h->host_only = 1;
  h->vsrc = mp3; h->vfed = 0;
  size_t fed = 0;
  unsigned main_top = 0;
  uint64_t md_end = 0;
  int sky_n = 0, next = 1, rc = -1;
  long long f = 0, ws0[4] = {-1, -1, -1, -1}, ws1[4] = {-1, -1, -1, -1};
  int part = 0;                                         /* whose records: 0 = none, the stream itself */
  long long part_i = 0;
  long long part_end = P->J > 1 ? seg_wait_start(&P->seg[1]) : -3;     /* where part 0 ends */
  if (part_end == -1) goto out;
  for (;;) {
    if (P->abort || P->quit) goto out;                  /* (given up, or the stitcher has left: rc stays "not regular") */
    if (next < P->snap_cap && f == (long long)next * P->sub) {        /* window `next` starts here */
      span_snap* S = &P->snap[next];
      S->frame = f; S->istart = h->istart; S->iend = h->iend; S->processed = h->processed; S->vfed = h->vfed; S->fed = fed;
      S->main_top = main_top; S->sky_n = sky_n;
      S->sky = (int*)malloc((size_t)(sky_n + 1) * sizeof(int));
      if (!S->sky) goto out;
      memcpy(S->sky, sky, (size_t)sky_n * sizeof(int));
      memcpy(S->last_ws0, ws0, sizeof ws0); memcpy(S->last_ws1, ws1, sizeof ws1);
      S->ready = 1;
      __atomic_store_n(&P->published, (long long)next + 1, __ATOMIC_RELEASE);
      next++;
    }
    while (ring_filled(h) < HOP_END_BYTES) {            /* H10 + the CLI's feeds (bulk_drive) */
      const size_t take = n - fed < 4096 ? n - fed : 4096;
      if (!take) { rc = 0; goto out; }                  /* the stream ends here: what is left is dropped, as the CLI drops it */
      if (take > ring_free_logical(h)) goto out;        /* a feed the CLI would drop (H16) */
      if (pdmp3_feed(h, mp3 + fed, take) != PDMP3_OK) goto out;
      fed += take;
    }
    const size_t x = h->vfed - ring_filled(h);
    const unsigned avail = ring_filled(h);
    hop1 own;
    const hop1* q = &own;
    if (part == 0 && part_end >= 0 && x >= (size_t)part_end) {         /* part 0 is through: the hop threads' records from here */
      if (x != (size_t)part_end) goto out;              /* (the guess was not a frame boundary) */
      part = 1; part_i = 0;
      P->t_pre_part0 = now_s() - P->t0;
    }
    if (part == 0) {
      if (hop_parse(mp3, x, &own) != 0) goto out;
      hop_prefetch(mp3, x, own.fb);
    } else {
      for (;;) {
        pre_seg* G = &P->seg[part];
        /* (the records were written by another core, most of them a while ago: they come from its cache or from memory,
         *  a line per four frames -- asked for well ahead, or the loop runs at one such miss per line) */
        if (part_i < __atomic_load_n(&G->count, __ATOMIC_ACQUIRE)) { __builtin_prefetch(&G->rec[part_i + 64], 0, 3); q = &G->rec[part_i++]; break; }
        const int st = __atomic_load_n(&G->state, __ATOMIC_ACQUIRE);
        if (st == SEG_RUNNING) { if (P->abort || P->quit) goto out; const double tw = now_s(); sched_yield(); P->t_pre_wait += now_s() - tw; continue; }
        if (part_i < __atomic_load_n(&G->count, __ATOMIC_ACQUIRE)) continue;   /* (its last records came with the state) */
        if (st != SEG_AT_NEXT || part + 1 >= P->J) goto out;           /* a header the regular path does not take, or a guess that was none */
        part++; part_i = 0;
      }
      if (q->x != x) goto out;
    }
    const unsigned nch = q->nch, nbytes = nch == 1 ? 17 : 32, crc = q->crc, fb = q->fb, begin = q->begin, top = q->top;
    if (fb > avail) goto out;                           /* (a frame the ring does not hold completely: H18) */
    if (begin > main_top) goto out;                     /* reservoir underflow (H9) */
    if (f >= P->rec_cap) goto out;
    hop_rec* r = &P->rec[f];
    md_end += top - begin;
    r->x = (uint32_t)x; r->md_src = (uint32_t)(x + 4 + crc + nbytes); r->md_end = md_end;
    r->fb = (uint16_t)fb; r->begin = (uint16_t)begin; r->top = (uint16_t)top; r->nch = (uint8_t)nch; r->crc = (uint8_t)crc;
    main_top = top;
    while (sky_n && P->rec[sky[sky_n - 1]].top <= main_top) sky_n--;
    sky[sky_n++] = (int)f;
    {                                                   /* (no branches on the stream's bits: they do not predict) */
      const unsigned wsb = q->ws, live = nch == 2 ? 0xfu : 0x5u;
      for (unsigned g = 0; g < 4; g++) {
        const long long on = -(long long)((wsb >> g) & (live >> g) & 1u), off = -(long long)((~wsb >> g) & (live >> g) & 1u);
        ws1[g] = (ws1[g] & ~on) | (f & on);
        ws0[g] = (ws0[g] & ~off) | (f & off);
      }
    }
    h->istart = (h->istart + fb) % INBUF_SIZE;
    h->processed += fb;
    h->l_istart = h->istart; h->l_processed = h->processed;
    if (h->processed > fed) goto out;                   /* (a replayed ring) */
    f++;
  }
out:
  free(h); free(sky);
  pthread_mutex_lock(&P->mu);
  P->n_frames = f;
  P->irregular = rc != 0;
  __atomic_store_n(&P->prepass_done, 1, __ATOMIC_RELEASE);
  pthread_cond_broadcast(&P->cv);
  pthread_mutex_unlock(&P->mu);
  return rc;
}
static void* par_prepass_thread(void* arg) {
  struct par_scan* P = (struct par_scan*)arg;
  const double t0 = now_s();
  (void)par_prepass(P);
  const double dt = now_s() - t0;
  pthread_mutex_lock(&P->mu); P->t_prepass = dt; pthread_mutex_unlock(&P->mu);     /* (read by the stitcher's trace) */
  return NULL;
}

/* `len` bytes of the stream's main data, from position `off` of their concatenation, whose last byte belongs to frame `g` or an earlier one */
static void md_read(const struct par_scan* P, long long g, uint64_t off, unsigned len, uint8_t* out) {
  while (g > 0 && P->rec[g - 1].md_end > off) g--;      /* the frame that holds byte `off` */
  while (len) {
    const hop_rec* r = &P->rec[g];
    const uint64_t start = r->md_end - (uint64_t)(r->top - r->begin);
    const unsigned in = (unsigned)(off - start), have = (unsigned)(r->md_end - off);
    const unsigned k = have < len ? have : len;
    memcpy(out, P->mp3 + r->md_src + in, k);
    out += k; off += k; len -= k; g++;
  }
}

/* a scanner's handle as the sequential scanner's would be in front of frame S->frame */
static void span_init(const struct par_scan* P, const span_snap* S, pdmp3_handle* id) {
  id->vsrc = P->mp3; id->vfed = S->vfed;
  id->istart = S->istart; id->iend = S->iend; id->processed = S->processed;
  id->l_istart = S->istart; id->l_processed = S->processed;
  id->new_header = 1; id->l_new_header = 1; id->need_reset = 0; id->ostart = 0;
  const hop_rec* last = &P->rec[S->frame - 1];
  header_fields(be32(P->mp3 + last->x), &id->hdr);
  id->l_hdr = id->hdr;
  id->last_nch = last->nch;
  /* the reservoir buffer: [0, top) of the newest frame, above it what older frames with larger tops left (sky), zero
   * where no frame ever reached */
  id->main_top = S->main_top;
  memset(id->main_vec, 0, sizeof id->main_vec);
  unsigned covered = 0;
  for (int i = S->sky_n - 1; i >= 0; i--) {
    const long long g = S->sky[i];
    const hop_rec* r = &P->rec[g];
    if (r->top <= covered) continue;
    md_read(P, g, r->md_end - r->top + covered, r->top - covered, id->main_vec + covered);
    covered = r->top;
  }
  /* side-info fields that a frame only sets on one side of win_switch_flag and otherwise leaves as they were (H20) */
  for (unsigned g = 0; g < 4; g++) {
    const unsigned gr = g >> 1, ch = g & 1;
    for (int which = 0; which < 2; which++) {
      const long long f = which ? S->last_ws1[g] : S->last_ws0[g];
      if (f < 0) continue;
      const hop_rec* r = &P->rec[f];
      const uint8_t* v = P->mp3 + r->x + 4 + r->crc;
      const unsigned pos = (r->nch == 1 ? 18u : 20u) + 59u * (gr * r->nch + ch);
      uint8_t tmp[48];
      memcpy(tmp, v, 40); memset(tmp + 40, 0, 8);       /* (side_word reads 8 bytes at a time) */
      const uint64_t xw = side_word(tmp, pos);
      const unsigned y = (unsigned)(xw >> 8) & 0x3fffff;
      if (which) for (unsigned w = 0; w < 3; w++) id->si.subblock_gain[gr][ch][w] = (y >> (6 - 3 * w)) & 7;
      else id->si.table_select[gr][ch][2] = (y >> 7) & 31;
    }
  }
}

typedef struct scanner_arg { struct par_scan* P; int k; int rc; } scanner_arg;
static void* par_scanner(void* arg) {
  scanner_arg* A = (scanner_arg*)arg;
  struct par_scan* P = A->P;
  A->rc = -1;
  par_cache* pc = P->b->pc;                             /* (scanner k's scratch is its own: no lock) */
  const size_t pool_bytes = (size_t)P->sub * RESERVOIR_BYTES + PDMP3_POOL_SLACK_BYTES + 64;
  if (!pc->scan[A->k].wb) pc->scan[A->k].wb = (struct bulk*)calloc(1, sizeof(struct bulk));
  if (!pc->scan[A->k].id) pc->scan[A->k].id = (pdmp3_handle*)malloc(sizeof(pdmp3_handle));
  if (pc->scan[A->k].pool_bytes < pool_bytes) {
    free(pc->scan[A->k].pool);
    pc->scan[A->k].pool = (uint8_t*)malloc(pool_bytes);
    pc->scan[A->k].pool_bytes = pc->scan[A->k].pool ? pool_bytes : 0;
  }
  struct bulk* wb = pc->scan[A->k].wb;
  pdmp3_handle* id = pc->scan[A->k].id;
  uint8_t* pool = pc->scan[A->k].pool;
  if (!wb || !id || !pool) goto done;
  const int whole = P->K == 1;                          /* one scanner: the sequential stage A, from frame 0 to the end */
  for (;;) {
    const long long w = __atomic_fetch_add(&P->next_win, 1, __ATOMIC_RELAXED);
    const double t_take = now_s();
    /* its snapshot: a few tens of microseconds away as a rule (the pre-pass leaves one every 10-25 us and the scanners take
     * them in turn).  The scanner that is next, or next but one, keeps looking; those further ahead sleep 20 us at a time.
     * (Not further than PAR_AHEAD windows in front of the stitcher either: finished windows are memory.) */
    int ready = 0;
    while (!P->abort && !P->quit) {
      const long long pub = __atomic_load_n(&P->published, __ATOMIC_ACQUIRE);
      if (w < pub) {
        if (w < __atomic_load_n(&P->stitched, __ATOMIC_RELAXED) + PAR_AHEAD) { ready = 1; break; }
      } else if (w >= P->snap_cap) break;
      else if (__atomic_load_n(&P->prepass_done, __ATOMIC_ACQUIRE)) {
        /* the stream ended (or went irregular) before this window -- unless its snapshot came with the end */
        if (w < __atomic_load_n(&P->published, __ATOMIC_ACQUIRE)) continue;
        break;
      }
      if (w >= pub && w < pub + 2) {
        if (P->spin) for (int i = 0; i < 64; i++) hp_pause(); else sched_yield();
      } else {
        const struct timespec nap = {0, 20000};
        (void)nanosleep(&nap, NULL);
      }
    }
    const int stop = P->abort;
    if (stop) break;
    if (!ready || (whole && w > 0)) { A->rc = 0; break; }
    memset(id, 0, sizeof *id);
    memset(wb, 0, sizeof *wb);
    id->host_only = 1;
    id->iso = P->b->id->iso;
    id->side_to_bits = 1; id->bits_scan = 1;
    id->pool_sink = wb;
    wb->id = id; wb->cap = P->sub; wb->bits_mode = 1; wb->pool_mode = 1; wb->win_sink = 1; wb->par = P; wb->pc = pc; wb->priv_pool = pool;
    wb->carry = P->b->carry;
    wb->stream_win = w;
    size_t fed = 0;
    if (w == 0) { pdmp3_open_feed(id); id->vsrc = P->mp3; id->vfed = 0; }
    else {
      const span_snap* S = &P->snap[w];
      span_init(P, S, id);
      fed = S->fed;
      wb->frames = S->frame;
    }
    wb->limit_frames = whole ? 0 : (w + 1) * P->sub;
    wb->tr_take = t_take; wb->tr_begin = now_s();
    size_t done;
    int res;
    while (!bulk_at_limit(wb) && (res = read_impl_sink(id, INBUF_SIZE, &done, wb)) != PDMP3_ERR) {
      if (P->abort || P->quit || wb->failed) break;
      if (id->processed > fed) break;                   /* (the pre-pass will have said so) */
      if (res == PDMP3_NEED_MORE) {
        const size_t take = P->n - fed < 4096 ? P->n - fed : 4096;
        if (!take) break;
        if (take > ring_free_logical(id)) break;
        (void)pdmp3_feed(id, P->mp3 + fed, take);
        fed += take;
      }
    }
    if (wb->failed || P->abort) break;
    if (P->quit) { A->rc = 0; break; }
    if (pw_close_window(wb) != PDMP3_OK) break;         /* (the window, full or -- the stream's last -- partly filled) */
  }
done:
  if (wb && wb->pw_cur) { pw_free(wb->pw_cur); wb->pw_cur = NULL; }
  pthread_mutex_lock(&P->mu);
  if (A->rc != 0) P->abort = 1;
  P->scanners_done++;
  pthread_cond_broadcast(&P->cv);
  pthread_mutex_unlock(&P->mu);
  return NULL;
}

/* Starts the hop threads, the pre-pass and the scanners for `mp3`; NULL when the stream is too short to bother or frame 0 is
 * not where a regular stream has it. */
#define PAR_MIN_WINDOWS 4                /* private windows (tests, a forced split scan); 12 otherwise */
#define PAR_MIN_PART_BYTES (1u << 18)   /* (a file of a few minutes -- 2 to 4 MB -- in six parts: the hop is a chain of cache misses, 0.1 us a frame on a stream that is not in the caches) */
static void par_free(struct par_scan* P) {          /* (the records and the hop threads' arrays are the cache's) */
  if (P->snap) for (long long w = 0; w < P->snap_cap; w++) free(P->snap[w].sky);
  free(P->win); free(P->snap); free(P->args); free(P);
}
static struct par_scan* par_start(struct bulk* b, const unsigned char* mp3, size_t n, int K, int sub, int min_windows, int max_parts) {
  if (K < 1 || n < 4096 || n > 0xfff00000u || (mp3[0] != 0xff) || (mp3[1] & 0xf0) != 0xf0) return NULL;
  frame_header H;
  header_fields(be32(mp3), &H);
  if (H.id != 1 || H.bitrate_index == 0 || H.bitrate_index == 15 || H.sfreq == 3 || H.layer != 3) return NULL;
  const unsigned fb0 = frame_bytes(&H);
  const long long est = (long long)(n / fb0);
  /* sub = 0: the caller leaves the private windows' size to the decoder: 256 frames.  (Until round 5: 1024 for long streams, 512
   * and 256 for files of a few minutes.  The host side takes as long either way; what the shorter ones buy is at the stream's
   * start -- the first window is with the stitcher after 0.05 ms instead of 0.13 -- and in how evenly the engine's windows fill:
   * 137812 frames with the PCM left in HBM, five runs each, interleaved: 31.6 M frames/s with 1024, 32.8 with 512, 34.2 with 256.) */
  if (sub == 0) sub = 256;
  if (sub > b->cap) sub = b->cap;
  if (sub < 1) return NULL;
  const long long est_windows = (est + sub - 1) / sub;
  if (est_windows < min_windows) return NULL;
  if (K > PAR_MAX_SCANNERS) K = PAR_MAX_SCANNERS;
  if (K > est_windows) K = (int)est_windows;
  /* parts of the pre-pass: $PDMP3_BULK_PREPASS_THREADS, else by the host's cores; none shorter than a megabyte */
  const int cores = usable_cpus();
  int J = cores >= 16 ? 6 : cores >= 12 ? 3 : 1;
  const char* ev = getenv("PDMP3_BULK_PREPASS_THREADS");
  size_t min_part = PAR_MIN_PART_BYTES;
  if (ev && atoi(ev) >= 1) { J = atoi(ev); min_part = 16384; }   /* (forced: tests split short streams) */
  if (K == 1) J = 1;                                    /* (one scanner: the sequential stage A, nothing to hurry for) */
  if (J > PAR_MAX_SEGS) J = PAR_MAX_SEGS;
  if (J > max_parts) J = max_parts;
  /* The parts grow: 1 : 1 : 1.5 : 2.25 : ... of the stream.  The pre-pass walks part 0 itself, at the hop's own speed (a chain
   * of cache misses: 30 ns a frame on a stream that is not in the caches), and reads the hop threads' records from there on at
   * 12 ns a frame -- but only as far as they have got: behind a part 0 of a 24th, five EQUAL parts had it follow the first hop
   * thread at that thread's pace through a fifth of the stream (the scanners, and the GPU behind them, waiting for snapshots for
   * the first third of the decode).  A part that is half as long again as the one before is through when the pre-pass gets there.
   * Measured, six interleaved runs each: the hour with the PCM left in HBM 34.5 against 34.4 M frames/s (the pre-pass's own 14 ns a
   * frame bound it either way), the corpus of files of a few minutes with one decoder 9.2 against 7.6 M (a file's part 0 is a
   * fourteenth of it instead of a quarter MB).  PDMP3_BULK_PREPASS_EQUAL=1: the old division. */
  double pw[PAR_MAX_SEGS], pw_sum = 0;
  const char* eq = getenv("PDMP3_BULK_PREPASS_EQUAL");
  const int equal_parts = eq && *eq == '1';
  for (;;) {
    pw_sum = 0;
    for (int j = 0; j < J; j++) { pw[j] = equal_parts ? (j == 0 ? 1.0 : 23.0 / (J > 1 ? J - 1 : 1)) : j < 2 ? 1.0 : pw[j - 1] * 1.5; pw_sum += pw[j]; }
    if (J == 1 || (double)n * pw[0] / pw_sum >= (double)min_part) break;
    J--;
  }
  size_t part_at[PAR_MAX_SEGS + 1];
  { double acc = 0; for (int j = 0; j < J; j++) { part_at[j] = (size_t)((double)n * acc / pw_sum); acc += pw[j]; } part_at[J] = n; }
  struct par_scan* P = (struct par_scan*)calloc(1, sizeof *P);
  if (!P) return NULL;
  P->b = b; P->mp3 = mp3; P->n = n; P->K = K; P->J = J; P->sub = sub; P->t0 = now_s();
  { const char* sp = getenv("PDMP3_BULK_SCAN_SPIN"); P->spin = sp ? atoi(sp) != 0 : cores >= 2 * K + 8; }
  P->published = 1;
  P->one_window = est + est / 16 <= b->cap;
  if (!b->pc) {
    if (!(b->pc = pc_new())) { free(P); return NULL; }
    b->pc->near_gpu = b->near_gpu;
  }
  par_cache* pc = b->pc;
  P->rec_cap = (long long)(n / 96) + 8;                 /* (no Layer III frame is shorter than 96 bytes) */
  if (pc->rec_cap < P->rec_cap) {
    free(pc->rec);
    pc->rec = (hop_rec*)malloc((size_t)P->rec_cap * sizeof(hop_rec));
    pc->rec_cap = pc->rec ? P->rec_cap : 0;
  }
  P->rec = pc->rec;
  P->win_cap = P->rec_cap / sub + 8;
  P->win = (pre_window**)calloc((size_t)P->win_cap, sizeof(pre_window*));
  P->snap_cap = P->win_cap;
  P->snap = (span_snap*)calloc((size_t)P->snap_cap, sizeof(span_snap));
  P->args = (scanner_arg*)calloc((size_t)K, sizeof(scanner_arg));
  int ok = P->rec && P->win && P->snap && P->args;
  for (int j = 1; ok && j < J; j++) {
    pre_seg* S = &P->seg[j];
    const size_t share = part_at[j + 1] - part_at[j];
    S->P = P; S->j = j; S->guess = part_at[j]; S->x_start = -2;
    S->cap = (long long)((share + 65536 + 4096) / 96) + 8;
    if (pc->seg_cap[j] < S->cap) {
      free(pc->seg_rec[j]);
      pc->seg_rec[j] = (hop1*)malloc((size_t)S->cap * sizeof(hop1));
      pc->seg_cap[j] = pc->seg_rec[j] ? S->cap : 0;
    }
    S->rec = pc->seg_rec[j];
    if (!S->rec) ok = 0;
  }
  if (!ok) { par_free(P); return NULL; }
  pthread_mutex_init(&P->mu, NULL); pthread_cond_init(&P->cv, NULL);
  /* (a thread that cannot be had: the ones that run are told to stop and waited for, and the stream goes the one-thread way) */
  int started = 1;
  for (int j = 1; started && j < J; j++) started = crew_run(pc, par_hop_thread, &P->seg[j], &P->hops_left) == 0;
  if (started) started = crew_run(pc, par_prepass_thread, P, &P->jobs_left) == 0;
  for (int k = 0; started && k < K; k++) {
    P->args[k].P = P; P->args[k].k = k; P->args[k].rc = -1;
    started = crew_run(pc, par_scanner, &P->args[k], &P->jobs_left) == 0;
  }
  crew_kick(pc);
  if (!started) {
    pthread_mutex_lock(&P->mu); P->abort = 1; pthread_cond_broadcast(&P->cv); pthread_mutex_unlock(&P->mu);
    crew_wait(pc, &P->jobs_left); crew_wait(pc, &P->hops_left);
    pthread_mutex_destroy(&P->mu); pthread_cond_destroy(&P->cv);
    par_free(P);
    return NULL;
  }
  return P;
}
/* joins the threads and frees everything; returns the pre-pass's verdict: 0 = the stream was regular and complete */
static int par_finish(struct par_scan* P) {
  par_cache* pc = P->b->pc;
  pthread_mutex_lock(&P->mu); P->quit = 1; pthread_cond_broadcast(&P->cv); pthread_mutex_unlock(&P->mu);
  /* (the pre-pass -- its loop and its waits for the hop threads -- and the scanners leave on `quit` or `abort`; the hop
   * threads look at abort only -- they are through long before unless the stitcher gave up early: told to stop once the
   * verdict is taken, which the others' end no longer changes.  A pre-pass that left on `quit` before its end reports
   * "not regular": the verdict below is then -1, as for any stream the stitcher did not see to its end) */
  crew_wait(pc, &P->jobs_left);
  pthread_mutex_lock(&P->mu); const int ok = !P->irregular && !P->abort; P->abort = 1; pthread_cond_broadcast(&P->cv); pthread_mutex_unlock(&P->mu);
  crew_wait(pc, &P->hops_left);
  for (long long w = 0; w < P->win_cap; w++) pw_free(P->win[w]);
  pthread_mutex_destroy(&P->mu); pthread_cond_destroy(&P->cv);
  par_free(P);
  return ok ? 0 : -1;
}
/* next finished window in stream order, or NULL: the stream is complete (*end = 1) or the scan was given up (*end = -1) */
static pre_window* par_next_window_wait(struct par_scan* P, long long w, int* end, double wait_s) {
  pre_window* pw = NULL;
  *end = 0;
  pthread_mutex_lock(&P->mu);
  for (;;) {
    if (P->abort || (P->prepass_done && P->irregular)) { *end = -1; break; }
    if (w < P->win_cap && P->win[w]) { pw = P->win[w]; P->win[w] = NULL; __atomic_store_n(&P->stitched, w + 1, __ATOMIC_RELAXED); break; }
    if (P->prepass_done && w >= (P->n_frames + P->sub - 1) / P->sub) { *end = 1; break; }
    if (P->scanners_done == P->K && P->prepass_done) { *end = -1; break; }      /* (a window is missing: should not happen) */
    if (wait_s < 0) { pthread_cond_wait(&P->cv, &P->mu); continue; }
    if (wait_s == 0) break;                             /* (not there yet: *end stays 0) */
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    ts.tv_nsec += (long)(wait_s * 1e9);
    if (ts.tv_nsec >= 1000000000L) { ts.tv_sec++; ts.tv_nsec -= 1000000000L; }
    (void)pthread_cond_timedwait(&P->cv, &P->mu, &ts);
    wait_s = 0;                                         /* (one more look, then back to the caller) */
  }
  pthread_mutex_unlock(&P->mu);
  return pw;
}
static pre_window* par_next_window(struct par_scan* P, long long w, int* end) { return par_next_window_wait(P, w, end, -1.0); }

static void par_trace_prepass(const struct par_scan* P) {
  fprintf(stderr, "  pre-pass in %d parts: its own part done at %.2f ms, waited %.2f ms for hop threads, through at %.2f ms;", P->J, P->t_pre_part0 * 1e3,
          P->t_pre_wait * 1e3, P->t_prepass * 1e3);
  for (int j = 1; j < P->J; j++) fprintf(stderr, " hop %d: start found %.2f, done %.2f (%lld frames);", j, P->seg[j].t_sync * 1e3, P->seg[j].t_done * 1e3, P->seg[j].count);
  fprintf(stderr, "\n");
}
static void pw_trace(const pre_window* pw, long long w, double t_start, double t0, double t1) {
  fprintf(stderr, "  window %lld: %d frames, taken %.2f, snapshot there %.2f, scanned %.2f, stitcher asked %.2f, got it %.2f ms\n", w, pw->n,
          (pw->t_take - t_start) * 1e3, (pw->t_begin - t_start) * 1e3, (pw->t_done - t_start) * 1e3, (t0 - t_start) * 1e3, (t1 - t_start) * 1e3);
}

/* windows of the engine that are still the GPU's (or the submitter's: handed over, not yet launched) */
static int bulk_in_flight(struct bulk* b) {
  int n = 0;
  for (int slot = 0; slot < BULK_SLOTS; slot++) {
    const bulk_flight* f = &b->flight[slot];
    if (!f->active) continue;
    pthread_mutex_lock(&b->sub_mu);
    const int launched = b->sub_tail > f->sub_seq;
    pthread_mutex_unlock(&b->sub_mu);
    if (!launched || pdmp3_hip_stream_done(b->hs, slot) == 0) n++;
  }
  return n;
}
/* a private window onto the end of the engine's open window: its pool behind what is there, offsets moved accordingly */
static int par_fits(const struct bulk* b, const pre_window* pw, int lim) {
  return (b->bits_n + pw->n <= lim || (!b->bits_n && pw->n <= b->cap)) && b->pool_tail + pw->pool_tail + POOL_ROOM <= b->pool_cap &&
         b->gath_n + pw->gath_n <= b->cap + BULK_GATH_EXTRA * (PAR_MAX_BATCH + 1) && b->slot_arena_n[b->bits_slot] < PAR_MAX_BATCH;
}
static void par_append(struct bulk* b, pre_window* pw) {
  const int at = b->bits_n;
  const uint32_t base = (uint32_t)b->pool_tail;
  memcpy(b->bits_dst + at, pw->bits, (size_t)pw->n * sizeof(pdmp3_frame_bits));
  memcpy(b->flight[b->bits_slot].nch + at, pw->nch, (size_t)pw->n);
  pdmp3_row_desc* d = b->desc_dst + at;
  for (int i = 0; i < pw->n; i++) { d[i] = pw->desc[i]; d[i].row_off += base; d[i].s_off += base; }
  struct pool_copy* g = b->gath_cur + b->gath_n;
  const struct pool_copy* gs = (const struct pool_copy*)pw->gath;
  for (int i = 0; i < pw->gath_n; i++) { g[i] = gs[i]; g[i].dst += base; }
  gather_push(b, b->bits_slot, b->res_dst, g, pw->gath_n);       /* (under way while the window fills) */
  b->g_pushed[b->bits_slot] = 1;
  b->bits_n += pw->n; b->gath_n += pw->gath_n; b->pool_tail += pw->pool_tail;
  b->slot_arena[b->bits_slot][b->slot_arena_n[b->bits_slot]++] = pw->arena; pw->arena = NULL;   /* (the copy list points into it until the submitter is through) */
}

/* The whole-stream decoder's stage A on several threads.  Returns the PCM byte count like bulk_drive, or -3: the stream is
 * not one the split scan takes (nothing has been changed), or -4: it was given up half way (windows of the stream's start
 * may have gone to the engine: the caller drains the pipeline and decodes the stream again the sequential way -- same
 * PCM for the frames both saw, so nothing wrong is ever left in the caller's buffer).
 *
 * The scanners' private windows are SHORT (`sub` frames: the first is there after 0.1 ms) and the engine's windows are
 * made of as many of them as there are when a slot is free, up to the slot's capacity: the GPU has something to do at
 * once, and once it is busy the windows grow by themselves to the size at which the device's Huffman stage fills the
 * chip (k_unpack: one workgroup per 16 frames, two per CU -- 8192 frames).  While two windows or more are still the
 * GPU's, a window that is not full waits for more. */
static atomic_int g_par_active;                    /* split scans under way in this process */
long long par_drive(struct bulk* b, const unsigned char* mp3, size_t n, int K) {
  const double t_start = now_s();
  int sub = 0;                                          /* (by the stream's length: par_start) */
  const char* se = getenv("PDMP3_BULK_SUB_FRAMES");
  if (se && atoi(se) >= 1) sub = atoi(se) < b->cap ? atoi(se) : b->cap;
  /* streams from 12 private windows on (3072 frames: a scan of 0.25 ms on one thread); a forced split scan: from 4 */
  /* several decoders of one process at it at once (a corpus, a decoder per few files): they share the host's cores --
   * the second takes half the scanners, the third and fourth a third and a quarter (two at least), and only the first
   * has hop threads */
  const int others = atomic_fetch_add(&g_par_active, 1);
  if (others > 0) { K = K / (others + 1); if (K < 2) K = 2; }
  struct par_scan* P = par_start(b, mp3, n, K, sub, b->scan_forced ? PAR_MIN_WINDOWS : 3 * PAR_MIN_WINDOWS, others > 0 ? 1 : PAR_MAX_SEGS);
  if (!P) { atomic_fetch_sub(&g_par_active, 1); return PAR_NOT_TAKEN; }
  sub = P->sub;
  K = P->K;
  long long total = 0, frames = 0, w = 0;
  int end = 0, engine_ok = 1, n_windows = 0, gave_up = 0;
  double t_win = 0, t_open = 0, t_fill = 0, t_more = 0;
  const char* tr = getenv("PDMP3_BULK_TRACE");
  const int trace2 = tr && atoi(tr) >= 2;
  b->trace2 = trace2; b->tr_t0 = t_start;
  pre_window* held = NULL;                              /* taken from the scanners, did not fit the window before */
  for (;;) {
    const double t0 = now_s();
    const int was_held = held != NULL;                  /* (taken from the scanners -- and counted -- when it did not fit the window before) */
    pre_window* pw = held ? held : par_next_window(P, w, &end);
    held = NULL;
    const double t1 = now_s();
    t_win += t1 - t0;
    if (!pw) break;
    if (!was_held) {
      if (trace2) pw_trace(pw, w, t_start, t0, t1);
      w++;
    }
    const int opened = engine_ok && bits_open_window(b) == PDMP3_OK;
    const double t2 = now_s();
    t_open += t2 - t1;
    /* (Tried in round 5: a short LAST window -- the stream's end known from the pre-pass, the window that would leave less than
     *  2048 frames behind stopping that far short of it -- so that the caller waits for a shorter chain of kernels at the end:
     *  32.2 against 33.8 M frames/s without, five interleaved runs.  The GPU is the bound by then, and two windows cost it more
     *  than one.  And capped FIRST windows -- cap / 8, cap / 8, cap / 4, cap / 2, as the one-thread scan has them -- so that the
     *  third window does not wait until it is full while two tiny ones are the GPU's: 33.4 against 33.7 M, six runs each.) */
    /* (a destination in host memory: windows of `target` frames -- 4096 unless the caller named a size -- or, for a stream that
     *  fits a slot, the slot: bulk_decode_impl's cur_target) */
    const int lim = b->pcm_pinned == 2 || b->cur_target <= 0 || b->cur_target > b->cap ? b->cap : b->cur_target;
    if (opened && par_fits(b, pw, lim)) {
      par_append(b, pw);
      for (int i = 0; i < pw->n; i++) total += 2304 * pw->nch[i];
      frames += pw->n;
      pw_free(pw);
      const double t3 = now_s();
      t_fill += t3 - t2;
      while (b->bits_n < lim) {                         /* what else is there, or worth waiting for */
        int e2;
        pre_window* more = par_next_window_wait(P, w, &e2, 0);
        /* (a stream that fits one slot goes up as ONE window -- a file of a few minutes: every window costs the GPU its
         *  150 us whatever it holds, and in a corpus the GPU has the file before to work on meanwhile) */
        const int hold = e2 == 0 && !more && (P->one_window || bulk_in_flight(b) >= 2);
        if (hold) more = par_next_window_wait(P, w, &e2, 50e-6);
        if (!more) { if (e2 == 0 && (P->one_window || bulk_in_flight(b) >= 2)) continue; break; }
        if (trace2) pw_trace(more, w, t_start, t3, now_s());
        w++;
        if (!par_fits(b, more, lim)) { held = more; break; }
        par_append(b, more);
        for (int i = 0; i < more->n; i++) total += 2304 * more->nch[i];
        frames += more->n;
        pw_free(more);
      }
      const double t4 = now_s();
      t_more += t4 - t3;
      b->frames = frames;
      if (trace2) fprintf(stderr, "  -> window of %d frames to slot %d at %.2f ms\n", b->bits_n, b->bits_slot, (t4 - t_start) * 1e3);
      n_windows++;
      if (bits_close_window(b) != PDMP3_OK) engine_ok = 0;
      t_fill += now_s() - t4;
    } else if (opened) {
      /* a private window that does not fit an EMPTY window of the engine (slots of a handful of frames: the window's pool has
       * no room for the reservoir image a private window starts with): not a stream for the split scan -- the one-thread scan
       * decodes it again from its first frame (PAR_GIVEN_UP), nothing is wrong with the engine */
      pw_free(pw);
      gave_up = 1;
      pthread_mutex_lock(&P->mu); P->abort = 1; pthread_cond_broadcast(&P->cv); pthread_mutex_unlock(&P->mu);
      break;
    } else { engine_ok = 0; pw_free(pw); }
    if (!engine_ok) { pthread_mutex_lock(&P->mu); P->abort = 1; pthread_cond_broadcast(&P->cv); pthread_mutex_unlock(&P->mu); break; }
  }
  pw_free(held);
  pthread_mutex_lock(&P->mu); const double t_pre = P->t_prepass; pthread_mutex_unlock(&P->mu);
  const long long nf = P->n_frames;
  if (trace2) par_trace_prepass(P);
  uint32_t last_hw = 0;
  if (end == 1 && nf > 0) last_hw = be32(mp3 + P->rec[nf - 1].x);
  const int fin = par_finish(P);
  const int ok = fin == 0 && end == 1 && engine_ok && frames == nf;
  atomic_fetch_sub(&g_par_active, 1);
  if (!ok && tr) fprintf(stderr, "bulk trace: split scan given up: pre-pass verdict %d, end %d, engine %d, frames %lld of %lld, %d windows%s\n", fin, end, engine_ok, frames, nf, n_windows,
                         gave_up ? " (a private window larger than an empty window of the engine)" : "");
  if (!engine_ok) { b->failed = 1; return -1; }
  if (!ok) { b->par_given_up++; return PAR_GIVEN_UP; }
  b->par_taken++;
  if (nf > 0) { header_fields(last_hw, &b->id->hdr); b->id->l_hdr = b->id->hdr; }
  if (getenv("PDMP3_BULK_TRACE"))
    fprintf(stderr, "bulk trace: split scan, %d scanners, %lld frames in private windows of %d, %d windows to the engine, pre-pass %.2f ms; stitch %.2f ms = "
            "waiting for the first private window of each %.2f + for slots %.2f + for more of them %.2f + filling and closing %.2f (this stream)\n",
            K, nf, sub, n_windows, t_pre * 1e3, (now_s() - t_start) * 1e3, t_win * 1e3, t_open * 1e3, t_more * 1e3, t_fill * 1e3);
  return total;
}

/* Host tests (no engine): the split scan's windows of a stream, in order, as one byte string -- per window n, the copy
 * list's length, the pool's fill, the side-info records, the row descriptors, the channel counts and the copy list (an
 * entry whose source is the stream as its offset, one whose source is the window's arena as its bytes).  K = 1 is the
 * unchanged stage-A code on one scanner from frame 0; K > 1 must give the same string.  Returns its length, -3 / -4 like
 * par_drive, -1 when `out` is too small. */
/* (the hook's stand-in for a decoder lives as long as the process, as a decoder's memory does from stream to stream;
 * one caller at a time) */
static pthread_mutex_t g_hook_mu = PTHREAD_MUTEX_INITIALIZER;
static struct bulk* g_hook_b;
static struct bulk* hook_get(void) {
  pthread_mutex_lock(&g_hook_mu);
  if (!g_hook_b) {
    g_hook_b = (struct bulk*)calloc(1, sizeof *g_hook_b);
    if (g_hook_b && !(g_hook_b->id = (pdmp3_handle*)calloc(1, sizeof *g_hook_b->id))) { free(g_hook_b); g_hook_b = NULL; }
  }
  if (!g_hook_b) pthread_mutex_unlock(&g_hook_mu);
  return g_hook_b;
}
static void hook_put(struct bulk* b) { (void)b; pthread_mutex_unlock(&g_hook_mu); }
long long pdmp3_amd_test_split_scan(const unsigned char* mp3, size_t n, int window_frames, int K, unsigned iso,
                                    unsigned char* out, size_t out_cap, long long* frames) {
  pthread_once(&g_lut_once, build_luts);
  struct bulk* b = hook_get();
  if (!b) return -1;
  b->cap = b->target = window_frames > 0 ? window_frames : 2048;
  b->bits_mode = 1; b->pool_mode = 1;             /* (what the scanners' sinks are: the window schedule depends on it) */
  b->id->iso = iso;
  const double t_start = now_s();
  struct par_scan* P = par_start(b, mp3, n, K, b->cap, PAR_MIN_WINDOWS, PAR_MAX_SEGS);
  if (!P) { hook_put(b); return PAR_NOT_TAKEN; }
  size_t o = 0;
  int end = 0, fit = 1;
  long long nf = 0;
#define PUT(ptr, len) do { if (o + (len) <= out_cap) memcpy(out + o, (ptr), (len)); else fit = 0; o += (len); } while (0)
  const char* tr = getenv("PDMP3_BULK_TRACE");
  const int trace2 = tr && atoi(tr) >= 2;
  for (long long w = 0;; w++) {
    const double t0 = trace2 ? now_s() : 0;
    pre_window* pw = par_next_window(P, w, &end);
    if (!pw) break;
    if (trace2) pw_trace(pw, w, t_start, t0, now_s());
    const int32_t hd[2] = {pw->n, pw->gath_n};
    const uint64_t pt = pw->pool_tail;
    PUT(hd, sizeof hd); PUT(&pt, sizeof pt);
    PUT(pw->bits, (size_t)pw->n * sizeof(pdmp3_frame_bits)); PUT(pw->desc, (size_t)pw->n * sizeof(pdmp3_row_desc)); PUT(pw->nch, (size_t)pw->n);
    const struct pool_copy* g = (const struct pool_copy*)pw->gath;
    for (int i = 0; i < pw->gath_n; i++) {
      const int lit = !(g[i].src >= mp3 && g[i].src < mp3 + n);
      const uint32_t e[3] = {(uint32_t)lit, g[i].dst, g[i].n};
      PUT(e, sizeof e);
      if (lit) PUT(g[i].src, g[i].n);
      else { const uint64_t off = (uint64_t)(g[i].src - mp3); PUT(&off, sizeof off); }
    }
    nf += pw->n;
    pw_free(pw);
  }
#undef PUT
  const long long pf = P->n_frames;
  if (trace2) par_trace_prepass(P);
  const int ok = par_finish(P) == 0 && end == 1 && nf == pf;
  hook_put(b);
  if (frames) *frames = nf;
  if (!ok) return PAR_GIVEN_UP;
  return fit ? (long long)o : -1;
}

