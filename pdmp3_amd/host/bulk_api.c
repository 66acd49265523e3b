/* bulk_api.c -- libpdmp3.so: the whole-stream decoder's entry points (include/pdmp3_bulk.h): pdmp3_amd_bulk_new* / delete,
 * decode / decode_async / wait, the parse-only forms for host tests, scan_buffer, pinned PCM buffers.
 * See host_internal.h for the map of the library. */
#include "bulk_internal.h"

void pdmp3_amd_bulk_delete(struct bulk* b) {
  if (!b) return;
  if (b->lsf_alt) { pdmp3_amd_bulk_delete(b->lsf_alt); b->lsf_alt = NULL; }
  if (b->th) {
    pthread_mutex_lock(&b->mu);
    b->quit = 1;
    pthread_cond_broadcast(&b->cv_work);
    pthread_mutex_unlock(&b->mu);
    for (int i = 0; i < b->nth; i++) pthread_join(b->th[i], NULL);
    free(b->th);
    pthread_mutex_destroy(&b->mu); pthread_cond_destroy(&b->cv_work); pthread_cond_destroy(&b->cv_done);
  }
  if (b->sub_started) {
    pthread_mutex_lock(&b->sub_mu);
    b->sub_quit = 1;
    pthread_cond_signal(&b->sub_cv);
    pthread_mutex_unlock(&b->sub_mu);
    pthread_join(b->sub_th, NULL);
    pthread_mutex_destroy(&b->sub_mu); pthread_cond_destroy(&b->sub_cv); pthread_cond_destroy(&b->sub_done_cv);
    pthread_mutex_lock(&b->gh_mu);
    b->gh_quit = 1;
    pthread_cond_broadcast(&b->gh_cv);
    pthread_mutex_unlock(&b->gh_mu);
    for (int i = 0; i < b->gh_n; i++) pthread_join(b->gh_th[i], NULL);
    pthread_mutex_destroy(&b->gh_mu); pthread_cond_destroy(&b->gh_cv); pthread_cond_destroy(&b->gh_done_cv);
  }
  for (int i = 0; i < 2; i++) { free(b->win[i].jobs); free(b->win[i].outs); }
  for (int i = 0; i < BULK_SLOTS; i++) {
    free(b->flight[i].nch); free(b->gath[i]);
    for (int k = 0; k < b->slot_arena_n[i]; k++) free(b->slot_arena[i][k]);
  }
  if (b->hs) pdmp3_hip_stream_destroy(b->hs);
  pc_free(b->pc);
  free(b->id);
  free(b);
}

/* threads <= 0: one per usable CPU (at most 64); window_frames <= 0: 2048.  with_engine = 0 gives a
 * parse-only decoder (host tests on machines without a GPU). */
static struct bulk* bulk_new(int threads, int window_frames, int with_engine, int bits_mode, int device) {
  pthread_once(&g_lut_once, build_luts);
  const int window_arg = window_frames;
  if (threads <= 0) {
    const int c = usable_cpus();
    threads = c > 64 ? 64 : c;
    /* (the pool only copies PCM out: at most 4 threads -- the scan, the submitter and on a device destination the split scan's
     *  threads want cores too, and more copy threads take memory bandwidth from the scanning thread, which bounds the
     *  host-memory destinations: the hour to pageable memory, five interleaved runs, 3 / 4 / 6 / 8 threads: 9.0 / 9.0 / 8.2 /
     *  8.1 M frames/s; six was the default for a while in round 5) */
    if (bits_mode && threads > 4) threads = 4;
  }
  /* frames per GPU batch.  Bits mode: k_unpack is bound by the length of one lane's chain, not by throughput, and two of
   * its workgroups (16 frames each) fit a CU: 8192 frames fill the chip once, 68 -> 84 us against 4096.  The slots hold
   * 8192; the split scan (PCM left on the device) fills them as far as its scanners have got when a slot is free, the
   * one-thread scan (host destinations: the PCM's way home over PCIe is what bounds those, and a window's PCM leaves when
   * the window is done) closes its windows at 4096 -- measured, pinned / pageable: 10.6 / 9.5 M frames/s against 9.6 / 8.1
   * with windows of 8192.  A window size the caller names is both. */
  int target = window_frames;
  if (window_frames <= 0) { window_frames = bits_mode ? 8192 : 2048; target = bits_mode ? 4096 : 2048; }
  if (window_frames > 32768) window_frames = target = 32768;
  struct bulk* b = (struct bulk*)calloc(1, sizeof *b);
  if (!b) return NULL;
  b->cap = window_frames;
  b->target = target;
  b->device = device;
  b->window_arg = window_arg;
  {
    /* split scan (par_drive): 12 scanners where the process has 32 CPUs, 8 with 16 (they live for the few milliseconds of a
     * stream's scan; with the PCM left in HBM the scanners, the upload and the kernels all take about 0.2 ms per 8192 frames, and
     * four more keep the scan off the critical path: 33.6 against 31.5-33.5 M frames/s, less spread), fewer on smaller quotas, none below 6 CPUs; PDMP3_BULK_SCAN_THREADS = 0 .. PAR_MAX_SCANNERS (16) overrides (0: one thread, as before) */
    const int c = usable_cpus();
    const char* e = getenv("PDMP3_BULK_SCAN_THREADS");
    b->scan_threads = e ? atoi(e) : (c >= 32 ? 12 : c >= 16 ? 8 : c >= 12 ? 4 : c >= 6 ? 2 : 0);
    b->scan_forced = e != NULL;
    if (b->scan_threads < 0) b->scan_threads = 0;
    if (b->scan_threads > PAR_MAX_SCANNERS) b->scan_threads = PAR_MAX_SCANNERS;
  }
  b->bits_mode = bits_mode;
  b->id = (pdmp3_handle*)calloc(1, sizeof *b->id);
  if (!b->id) { free(b); return NULL; }
  b->id->host_only = 1;
  for (int i = 0; i < 2 && !bits_mode; i++) {
    b->win[i].jobs = (frame_job*)malloc((size_t)b->cap * sizeof(frame_job));
    b->win[i].outs = (main_out*)malloc((size_t)b->cap * sizeof(main_out));
    if (!b->win[i].jobs || !b->win[i].outs) { pdmp3_amd_bulk_delete(b); return NULL; }
  }
  if (with_engine) {
    pdmp3_hip_ctx* ctx = shared_ctx_on(device);
    int made = PDMP3_HIP_EINVAL;
    if (ctx) {
      /* the slots' pinned buffers: allocated (and first touched, by the pinning) from a thread on the GPU's node */
      cpu_set_t before;
      const int rebind = gpu_local_cpus(ctx, &b->near_gpu) > 0 && sched_getaffinity(0, sizeof before, &before) == 0;
      if (rebind) (void)sched_setaffinity(0, sizeof b->near_gpu, &b->near_gpu);
      /* (the scanners / hop threads / gather helpers are SIZED by the whole affinity mask and BOUND to the
       *  GPU's node: on a host whose node holds half of the usable CPUs that oversubscribes the node.  Sizing them by the
       *  node was tried in round 6 and not kept without a measurement on such a host: on the 16-CPU quota of the test boxes
       *  it would cut the scanners from 8 to 2.  PDMP3_BULK_SCAN_THREADS / _PREPASS_THREADS / _GATHER_THREADS size them.) */
      made = pdmp3_hip_stream_create_slots(ctx, b->cap, BULK_SLOTS, &b->hs);
      if (rebind) (void)sched_setaffinity(0, sizeof before, &before);
    }
    if (made != PDMP3_HIP_OK) {
      fprintf(stderr, "pdmp3: no MI355X transform engine: %s\n", pdmp3_hip_last_error());
      b->hs = NULL;
      pdmp3_amd_bulk_delete(b);
      return NULL;
    }
    for (int i = 0; i < BULK_SLOTS; i++) {
      b->flight[i].nch = (uint8_t*)malloc((size_t)b->cap);
      b->gath[i] = (struct pool_copy*)malloc(((size_t)b->cap + BULK_GATH_EXTRA * (PAR_MAX_BATCH + 1)) * sizeof(struct pool_copy));
      if (!b->flight[i].nch || !b->gath[i]) { pdmp3_amd_bulk_delete(b); return NULL; }
    }
    if (bits_mode) {
      const char* snap = getenv("PDMP3_BULK_SNAPSHOT_ROWS");          /* 1: the 2064-byte-per-frame form of the input */
      b->pool_mode = !(snap && *snap && *snap != '0');
      pthread_mutex_init(&b->sub_mu, NULL); pthread_cond_init(&b->sub_cv, NULL); pthread_cond_init(&b->sub_done_cv, NULL);
      if (pthread_create(&b->sub_th, NULL, bulk_submitter, b) != 0) { pdmp3_amd_bulk_delete(b); return NULL; }
      b->sub_started = 1;
      bind_thread(b->sub_th, &b->near_gpu);
      pthread_mutex_init(&b->gh_mu, NULL); pthread_cond_init(&b->gh_cv, NULL); pthread_cond_init(&b->gh_done_cv, NULL);
      {
        const char* ge = getenv("PDMP3_BULK_GATHER_THREADS");      /* helpers for the windows' main-data copies (0 .. 8) */
        int want = ge ? atoi(ge) : (b->scan_threads >= 8 ? 6 : b->scan_threads > 0 ? 3 : 0);   /* (4 instead of 6: the same with the PCM left in HBM, 9.0 against 9.7 M frames/s to pinned memory) */
        if (want > GATHER_MAX_HELPERS) want = GATHER_MAX_HELPERS;
        for (b->gh_n = 0; b->gh_n < want; b->gh_n++)
          if (pthread_create(&b->gh_th[b->gh_n], NULL, gather_helper, b) != 0) break;
          else bind_thread(b->gh_th[b->gh_n], &b->near_gpu);
      }
    }
  }
  pthread_mutex_init(&b->mu, NULL); pthread_cond_init(&b->cv_work, NULL); pthread_cond_init(&b->cv_done, NULL);
  b->th = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
  if (!b->th) { pdmp3_amd_bulk_delete(b); return NULL; }
  for (b->nth = 0; b->nth < threads; b->nth++)
    if (pthread_create(&b->th[b->nth], NULL, bulk_worker, b) != 0) break;
    else if (bits_mode) bind_thread(b->th[b->nth], &b->near_gpu);   /* (a host-Huffman pool is compute: it takes every socket) */
  if (b->nth == 0) { pdmp3_amd_bulk_delete(b); return NULL; }
  return b;
}

/* default: Huffman decoding on the device; PDMP3_BULK_HOST_HUFFMAN=1 (or _new_ex) keeps it on the host pool */
struct bulk* pdmp3_amd_bulk_new_on(int threads, int window_frames, int host_huffman, int device) {
  return bulk_new(threads, window_frames, 1, !host_huffman, device);
}
struct bulk* pdmp3_amd_bulk_new_ex(int threads, int window_frames, int host_huffman) {
  return bulk_new(threads, window_frames, 1, !host_huffman, default_device());
}
struct bulk* pdmp3_amd_bulk_new(int threads, int window_frames) {
  const char* e = getenv("PDMP3_BULK_HOST_HUFFMAN");
  return bulk_new(threads, window_frames, 1, !(e && *e && *e != '0'), default_device());
}
struct bulk* pdmp3_amd_bulk_new_parse_only(int threads, int window_frames) { return bulk_new(threads, window_frames, 0, 0, 0); }
struct bulk* pdmp3_amd_bulk_new_parse_bits(void) { return bulk_new(1, 1, 0, 1, 0); }
int pdmp3_amd_bulk_threads(const struct bulk* b) { return b ? b->nth : 0; }
/* streams this decoder's split scan (several scanner threads: device destinations, or PDMP3_BULK_SCAN_THREADS) decoded to
 * their end, and streams it gave up half way and decoded again with the one-thread scan (irregular ones: resync, tags,
 * truncation in the middle of the ring's cadence) -- same PCM either way; for tests and for whoever wonders about the rate */
void pdmp3_amd_bulk_split_scans(const struct bulk* b, long long* taken, long long* given_up) {
  if (taken) *taken = b ? b->par_taken : 0;
  if (given_up) *given_up = b ? b->par_given_up : 0;
}
/* the ISO-correct switches (include/pdmp3.h: pdmp3_amd_set_quirks) for the streams this decoder is given from now on */
int pdmp3_amd_bulk_set_quirks(struct bulk* b, unsigned iso_mask) { return b ? pdmp3_amd_set_quirks(b->id, iso_mask) : PDMP3_ERR; }

static void bulk_begin(struct bulk* b) {
  pdmp3_handle* id = b->id;
  /* a fresh handle per stream -- unless the caller is pdmp3(), which decodes all its files with ONE handle: parse
   * state left by the previous file shows in the next one, so it is kept (b->carry) */
  if (!b->carry) {
    const unsigned iso = id->iso;                 /* (a setting of the decoder, not parse state: pdmp3_amd_bulk_set_quirks) */
    memset(id, 0, sizeof *id);
    id->host_only = 1;
    id->iso = iso;
  }
  id->pool_sink = b->pool_mode ? b : NULL;
  id->side_to_bits = b->bits_mode && !getenv("PDMP3_BULK_SLOW_SIDE_INFO");
  id->bits_scan = b->bits_mode;
  id->lsf_seen = 0;
  /* (windows, flights, a running copy job: the pipeline keeps going across streams) */
  b->win[b->cur].n = 0;
  b->frames = 0; b->pcm_emitted = 0; b->count_only = 0; b->failed = 0;
  b->bits_open = 0; b->bits_n = 0;            /* (a failed submit stays failed: sub_rc is sticky) */
  b->stream_win = 0;
}

/* frames and PCM bytes pdmp3() would produce for this stream: stage A alone */
long long pdmp3_amd_scan_buffer_iso(const unsigned char* mp3, size_t n, unsigned iso_mask, long long* frames);
long long pdmp3_amd_scan_buffer(const unsigned char* mp3, size_t n, long long* frames) { return pdmp3_amd_scan_buffer_iso(mp3, n, 0, frames); }
/* ... with the switches a decoder was given (pdmp3_amd_bulk_set_quirks): only PDMP3_ISO_LSF changes what a scan counts */
long long pdmp3_amd_scan_buffer_iso(const unsigned char* mp3, size_t n, unsigned iso_mask, long long* frames) {
  pthread_once(&g_lut_once, build_luts);
  struct bulk b;
  memset(&b, 0, sizeof b);
  b.id = (pdmp3_handle*)calloc(1, sizeof *b.id);
  if (!b.id) return -1;
  b.id->host_only = 1;
  b.id->iso = iso_mask & (PDMP3_ISO_ALL | PDMP3_ISO_LSF);
  b.count_only = 1;
  const long long total = bulk_drive(&b, mp3 ? mp3 : (const unsigned char*)"", mp3 ? n : 0);
  if (frames) *frames = b.frames;
  free(b.id);
  return total;                                   /* PDMP3_BULK_REPLAY (-2) passes through */
}

/* everything submitted so far is decoded and its PCM in caller memory */
static int bulk_drain(struct bulk* b) {
  int ok = sub_drain(b) == PDMP3_OK;
  bulk_wait_b(b);
  for (int i = 0; i < BULK_SLOTS && ok; i++) {         /* what is still on the GPU: the pool copies it out */
    const unsigned char* src; unsigned char* dst; size_t nbytes;
    ok = bulk_collect(b, i, &src, &dst, &nbytes) == PDMP3_OK;
    if (ok && nbytes) { bulk_start_b(b, NULL, src, dst, nbytes); bulk_wait_b(b); }
  }
  bulk_wait_b(b);
  b->in_b = NULL;
  if (!ok) for (int i = 0; i < BULK_SLOTS; i++) { (void)pdmp3_hip_stream_wait(b->hs, i); b->flight[i].active = 0; }
  return ok ? PDMP3_OK : PDMP3_ERR;
}

/* scan + submit one stream; with drain = 0 its last windows may still be on their way when this returns */
static long long bulk_decode_impl(struct bulk* b, const unsigned char* mp3, size_t n, unsigned char* pcm, size_t pcm_cap,
                                  long* rate, int* channels, int drain) {
  if (!b || !b->hs || (!mp3 && n) || (!pcm && pcm_cap)) return -1;
  /* PDMP3_ISO_LSF on a decoder whose Huffman stage is on the device: that stage reads MPEG-1 side info only, so a stream
   * that opens with an MPEG-2 LSF / 2.5 header goes through a host-Huffman decoder this one keeps for the purpose
   * (same device, same threads and window, same switches; the call is synchronous then) */
  if (b->bits_mode && (b->id->iso & PDMP3_ISO_LSF) && n >= 4 && mp3[0] == 0xff && (mp3[1] & 0xe0) == 0xe0 && (mp3[1] & 0x18) != 0x18 && (mp3[1] & 0x18) != 0x08) {
    if (!b->lsf_alt) b->lsf_alt = bulk_new(b->nth, b->window_arg, 1, 0, b->device);
    if (!b->lsf_alt) return -1;
    if (bulk_drain(b) != PDMP3_OK) return -1;     /* (this decoder's own streams first: the PCM destinations may overlap) */
    b->lsf_alt->id->iso = b->id->iso;
    return bulk_decode_impl(b->lsf_alt, mp3, n, pcm, pcm_cap, rate, channels, 1);
  }
  bulk_begin(b);
  /* Device Huffman: nothing to reset on the host side -- the stream's first frame carries PDMP3_FR_RESET (synthesis
   * state) and, unless parse state is carried over (pdmp3()), PDMP3_FR_NEWSTREAM (scalefactors / count1), so
   * streams follow each other through the pipeline without a stop.  Host Huffman: the pipeline is idle here. */
  if (!b->bits_mode && !b->carry && pdmp3_hip_stream_reset(b->hs) != PDMP3_HIP_OK) return -1;
  b->pcm = pcm; b->pcm_cap = pcm_cap;
  b->pcm_pinned = pcm_cap ? pdmp3_hip_host_is_pinned(pcm, pcm_cap) : 0;      /* 1 pinned host memory, 2 device memory */
  const double t_in = now_s();
  long long total = PAR_NOT_TAKEN;
  /* short first windows (win_ramp) only for streams long enough to gain from them: a file of a few minutes would go up
   * in five windows instead of two, and a window costs the GPU 150 us whatever its size (C4 corpus to pageable memory:
   * 5.2 -> 4.0 M frames/s with the ramp on every file) */
  b->ramp_on = 0;
  b->cur_target = b->target;
  if (n >= 4 && mp3[0] == 0xff && (mp3[1] & 0xf0) == 0xf0) {
    frame_header H0;
    header_fields(((uint32_t)mp3[0] << 24) | ((uint32_t)mp3[1] << 16) | ((uint32_t)mp3[2] << 8) | mp3[3], &H0);
    if (H0.id == 1 && H0.layer == 3 && H0.bitrate_index != 0 && H0.bitrate_index != 15 && H0.sfreq != 3)
    {
      const long long est0 = (long long)(n / frame_bytes(&H0));
      b->ramp_on = est0 >= 8LL * b->target;
      /* a file of a few minutes that fits one slot goes up as one window instead of a full one and a remainder (4096 + 32
       * frames, say: the remainder costs the GPU as much as the full one) */
      if (b->bits_mode && est0 > b->target && est0 + est0 / 16 <= b->cap) b->cur_target = b->cap;
    }
  }
  /* The split scan: with the PCM left in device memory the scan is the bound without it (13 -> 18 M frames/s in round 4, 35 now).
   * Towards host memory the link bounds the pipeline -- 12.7 ms for the hour's PCM at 50 GB/s -- but the ONE-thread scan takes
   * 12.4 ms of its own beside it and loses to every disturbance (pinned 8.9-9.8 M frames/s, pageable 8.2-9.6 on one box): since
   * round 5 host destinations take the split scan as well, with four scanners and the engine's windows closed at 4096 frames (a
   * window's PCM leaves when the window is done): pinned 10.2-10.4, pageable 9.35-9.54, five interleaved runs.  (Round 4's split
   * scan, with its pre-pass on the scanners' mutex and eight scanners, LOST there: pinned 10.4 -> 8.3.)
   * (Decoders whose windows the caller made shorter than 1024 frames, and hosts with fewer than 12 usable CPUs -- two scanners by
   * default: measured nowhere --, keep the one-thread scan for host destinations.)
   * PDMP3_BULK_SCAN_THREADS=0: the one-thread scan everywhere. */
  if (b->bits_mode && b->pool_mode && !b->carry && b->scan_threads > 0 &&
      (b->pcm_pinned == 2 || b->scan_forced || (b->target >= 1024 && b->scan_threads >= 4))) {
    /* (host destinations: four scanners at most -- the link bounds the pipeline there, the scan only has to stay off its path) */
    total = par_drive(b, mp3, n, b->pcm_pinned == 2 || b->scan_forced || b->scan_threads < 4 ? b->scan_threads : 4);
    if (total == PAR_GIVEN_UP) {
      /* not a stream the split scan can take after all (something irregular further in): what has gone to the engine
       * is let through, then the stream is decoded again from its first frame by the one-thread scanner */
      (void)bulk_drain(b);
      bulk_begin(b);
      b->pcm = pcm; b->pcm_cap = pcm_cap;
    }
  }
  if (total == PAR_NOT_TAKEN || total == PAR_GIVEN_UP) total = bulk_drive(b, mp3, n);
  if (b->bits_mode && b->id->lsf_seen) {
    /* an LSF frame somewhere behind the stream's first bytes (an ID3 tag in front, junk, an MPEG-1 stream that goes on as LSF): what
     * has gone to the engine is let through and dropped, and the stream is decoded again by the host-Huffman decoder */
    b->id->lsf_seen = 0;
    (void)bits_close_window(b);
    (void)bulk_drain(b);
    b->failed = 0;
    if (!b->lsf_alt) b->lsf_alt = bulk_new(b->nth, b->window_arg, 1, 0, b->device);
    if (!b->lsf_alt) return -1;
    b->lsf_alt->id->iso = b->id->iso;
    return bulk_decode_impl(b->lsf_alt, mp3, n, pcm, pcm_cap, rate, channels, 1);
  }
  const double t_driven = now_s();
  b->t_drive += t_driven - t_in;
  int ok = !b->failed;
  if (b->bits_mode) ok = bits_close_window(b) == PDMP3_OK && ok;
  else {
    ok = ok && bulk_rotate(b) == PDMP3_OK;               /* the partly filled last window */
    ok = ok && bulk_finish_b(b) == PDMP3_OK;
  }
  if (drain || !b->bits_mode || !ok) ok = bulk_drain(b) == PDMP3_OK && ok;
  else if (b->pool_mode) ok = sub_drain_copied(b) == PDMP3_OK && ok;      /* the submitter has taken the main data out of `mp3` */
  b->t_tail += now_s() - t_driven;
  if (rate) *rate = (long)kLsfSampleRates[sfreq9(&b->id->hdr)];
  if (channels) *channels = b->id->hdr.mode == 3 ? 1 : 2;
  if (getenv("PDMP3_BULK_TRACE")) {
    fprintf(stderr, "bulk trace: scan loop %.2f ms (of it waiting for the submitter %.2f ms), tail %.2f ms (cumulative)\n",
            b->t_drive * 1e3, b->t_subwait * 1e3, b->t_tail * 1e3);
    fprintf(stderr, "bulk trace: submit %.2f ms, gpu wait %.2f ms, pool wait %.2f ms; submitter: copies %.2f ms, engine calls %.2f ms (cumulative)\n",
            b->t_submit * 1e3, b->t_gpuwait * 1e3, b->t_poolwait * 1e3, b->t_sub_gather * 1e3, b->t_sub_call * 1e3);
  }
  if (total == PDMP3_BULK_REPLAY) return PDMP3_BULK_REPLAY;
  return ok ? total : -1;
}

/* PCM buffers in pinned host memory: the GPU downloads each window straight into them, the pool has nothing to copy */
void* pdmp3_amd_pcm_alloc(size_t bytes) {
  void* p = NULL;
  return pdmp3_hip_host_alloc(bytes, &p) == PDMP3_HIP_OK ? p : NULL;
}
void pdmp3_amd_pcm_free(void* p) { pdmp3_hip_host_free(p); }

/* Decode a whole stream.  Returns the PCM byte count pdmp3() writes for it (the first min(that, pcm_cap)
 * bytes are in `pcm`), or -1 on an engine failure.  rate / channels: format of the last header seen. */
long long pdmp3_amd_bulk_decode(struct bulk* b, const unsigned char* mp3, size_t n, unsigned char* pcm, size_t pcm_cap,
                                long* rate, int* channels) {
  return bulk_decode_impl(b, mp3, n, pcm, pcm_cap, rate, channels, 1);
}

/* The same without waiting for the tail: returns as soon as the stream is scanned and its windows are queued
 * (`mp3` may be released then); `pcm` is complete after pdmp3_amd_bulk_wait().  The next stream's scan overlaps
 * the previous one's GPU work and copy-out -- for corpora of many files.  (Host-Huffman decoders wait anyway.) */
long long pdmp3_amd_bulk_decode_async(struct bulk* b, const unsigned char* mp3, size_t n, unsigned char* pcm, size_t pcm_cap,
                                      long* rate, int* channels) {
  return bulk_decode_impl(b, mp3, n, pcm, pcm_cap, rate, channels, 0);
}

int pdmp3_amd_bulk_wait(struct bulk* b) {
  if (!b || !b->hs) return -1;
  return bulk_drain(b) == PDMP3_OK ? 0 : -1;
}

/* Host stages A-C only: the records the engine would be given, into caller memory (cap_frames frames).
 * Returns the frame count, or -1 when they do not fit. */
long long pdmp3_amd_bulk_parse(struct bulk* b, const unsigned char* mp3, size_t n, int16_t* spectra, pdmp3_gc_side* side,
                               size_t cap_frames, long long* pcm_bytes) {
  if (!b || b->hs || b->bits_mode || (!mp3 && n)) return -1;
  bulk_begin(b);
  b->rec_spectra = spectra; b->rec_side = side; b->rec_cap = cap_frames;
  const long long total = bulk_drive(b, mp3, n);
  int ok = !b->failed && bulk_rotate(b) == PDMP3_OK;
  ok = ok && bulk_finish_b(b) == PDMP3_OK;
  bulk_wait_b(b);
  b->in_b = NULL;
  if (pcm_bytes) *pcm_bytes = total;
  if (total == PDMP3_BULK_REPLAY) return PDMP3_BULK_REPLAY;
  return ok ? b->frames : -1;
}

/* Stage A only, bits mode, into caller memory: what pdmp3_hip_stream_submit_bits would be given (host tests) */
long long pdmp3_amd_bulk_parse_bits(struct bulk* b, const unsigned char* mp3, size_t n, pdmp3_frame_bits* bits, uint8_t* res,
                                    size_t cap_frames, long long* pcm_bytes) {
  if (!b || b->hs || !b->bits_mode || (!mp3 && n)) return -1;
  bulk_begin(b);
  b->rec_bits = bits; b->rec_res = res; b->rec_cap = cap_frames;
  const long long total = bulk_drive(b, mp3, n);
  if (pcm_bytes) *pcm_bytes = total;
  if (total == PDMP3_BULK_REPLAY) return PDMP3_BULK_REPLAY;
  return b->failed ? -1 : b->frames;
}

/* the same in the compact form (include/pdmp3_hip.h: pdmp3_row_desc): side info, row descriptors and the pool of ONE
 * window that holds the whole stream (host tests of the pool rule against the snapshot rows) */
long long pdmp3_amd_bulk_parse_pool(struct bulk* b, const unsigned char* mp3, size_t n, pdmp3_frame_bits* bits,
                                    pdmp3_row_desc* desc, uint8_t* pool, size_t pool_cap, size_t cap_frames, size_t* pool_bytes) {
  if (!b || b->hs || !b->bits_mode || (!mp3 && n)) return -1;
  b->pool_mode = 1;
  bulk_begin(b);
  b->rec_bits = bits; b->rec_res = pool; b->rec_desc = desc; b->rec_pool_cap = pool_cap; b->rec_cap = cap_frames;
  const long long total = bulk_drive(b, mp3, n);
  if (b->bits_open) { pool_materialize(b); pool_gather(pool, b->gath_cur, b->gath_n); }
  if (pool_bytes) *pool_bytes = b->pool_tail;
  b->pool_mode = 0; b->id->pool_sink = NULL;
  if (total == PDMP3_BULK_REPLAY) return PDMP3_BULK_REPLAY;
  return b->failed ? -1 : b->frames;
}

