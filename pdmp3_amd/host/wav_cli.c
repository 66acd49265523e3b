/* wav_cli.c -- libpdmp3.so: pdmp3() -- the reference's CLI driver (P:2540-2589) -- with the raw sink (P:2236-2257) and
 * the .wav writer.  See host_internal.h for the map of the library. */
#include "bulk_internal.h"

/* ------------------------------------------------------------------------ */
/* CLI driver (P:2540-2589) with the raw sink (P:2236-2257)                   */
/* ------------------------------------------------------------------------ */
/* RIFF/WAVE header for interleaved PCM: 16-bit integer (format 1) or 32-bit float (format 3); data_bytes = 0xffffffff
 * when the length is not known yet (a pipe) */
static void wav_header(unsigned char h[44], long rate, int channels, int float32, uint32_t data_bytes) {
  const uint32_t bps = float32 ? 4 : 2, align = bps * (uint32_t)channels;
  const uint32_t riff = data_bytes == 0xffffffffu ? 0xffffffffu : data_bytes + 36;
#define LE32(p, v) ((p)[0] = (unsigned char)(v), (p)[1] = (unsigned char)((v) >> 8), (p)[2] = (unsigned char)((v) >> 16), (p)[3] = (unsigned char)((v) >> 24))
#define LE16(p, v) ((p)[0] = (unsigned char)(v), (p)[1] = (unsigned char)((v) >> 8))
  memcpy(h, "RIFF", 4); LE32(h + 4, riff); memcpy(h + 8, "WAVEfmt ", 8); LE32(h + 16, 16u);
  LE16(h + 20, float32 ? 3u : 1u); LE16(h + 22, (uint32_t)channels); LE32(h + 24, (uint32_t)rate);
  LE32(h + 28, (uint32_t)rate * align); LE16(h + 32, align); LE16(h + 34, bps * 8);
  memcpy(h + 36, "data", 4); LE32(h + 40, data_bytes);
#undef LE32
#undef LE16
}

/* include/pdmp3_bulk.h: a whole PCM buffer as a .wav file */
int pdmp3_amd_write_wav(const char* path, const void* pcm, size_t bytes, long rate, int channels, int float32) {
  if (!path || (!pcm && bytes) || rate <= 0 || channels < 1 || channels > 2 || bytes > 0xfffffff0u) return PDMP3_ERR;
  FILE* f = fopen(path, "wb");
  if (!f) return PDMP3_ERR;
  unsigned char h[44];
  wav_header(h, rate, channels, float32, (uint32_t)bytes);
  int ok = fwrite(h, 1, 44, f) == 44 && (bytes == 0 || fwrite(pcm, 1, bytes, f) == bytes);
  ok = (fclose(f) == 0) && ok;
  return ok ? PDMP3_OK : PDMP3_ERR;
}

/* The driver's sink.  Default: the reference's OUTPUT_RAW writer (P:2236-2257) -- "<first name>.raw", opened once for
 * the FIRST file name only, O_CREAT without O_TRUNC, "-" = stdout.  PDMP3_CLI_WAV=1: the same samples as
 * "<first name>.wav" (truncated, 44-byte header with the first stream's rate and channel count, sizes filled in when
 * the driver is done; to stdout with unknown-length sizes). */
static int g_out_fd = -2, g_out_wav = 0, g_out_hdr = 0;
static uint64_t g_out_bytes = 0;
static long g_out_rate = 44100;
static int g_out_ch = 2;

static void write_all(int fd, const unsigned char* data, size_t nbytes) {
  size_t off = 0;
  while (off < nbytes) {
    ssize_t w = write(fd, data + off, nbytes - off);
    if (w <= 0) { fputs("Unable to write raw data\n", stderr); exit(-1); }
    off += (size_t)w;
  }
}

static void write_raw(const char* filename, const unsigned char* data, size_t nbytes, long rate, int channels) {
  if (g_out_fd == -2) {
    const char* w = getenv("PDMP3_CLI_WAV");
    g_out_wav = w && *w && *w != '0';
    if (strcmp(filename, "-")) {
      char name[1024];
      snprintf(name, sizeof name, g_out_wav ? "%s.wav" : "%s.raw", filename);
      g_out_fd = open(name, O_WRONLY | O_CREAT | (g_out_wav ? O_TRUNC : 0), 0666);  /* raw: no O_TRUNC, like the reference */
      if (g_out_fd == -1) { perror(name); exit(-1); }
    } else g_out_fd = 1;
  }
  /* the sink is opened by the FIRST call, data or not -- the reference calls its writer after every pdmp3_read, the
   * first NEED_MORE with nothing decoded included (P:2565-2566, P:2239-2251), so the output is named after the first
   * file even if that one yields no PCM; only the WAV header waits for the first data (rate and channels) */
  if (g_out_wav && !g_out_hdr && nbytes) {
    unsigned char h[44];
    g_out_rate = rate > 0 ? rate : 44100; g_out_ch = channels == 1 ? 1 : 2;
    wav_header(h, g_out_rate, g_out_ch, 0, 0xffffffffu);
    write_all(g_out_fd, h, 44);
    g_out_hdr = 1;
  }
  write_all(g_out_fd, data, nbytes);
  g_out_bytes += nbytes;
}

static void finish_output(void) {
  if (g_out_fd >= 0 && g_out_wav && !g_out_hdr) {            /* no data at all: an empty WAV file */
    unsigned char h[44];
    wav_header(h, g_out_rate, g_out_ch, 0, 0);
    write_all(g_out_fd, h, 44);
    g_out_hdr = 1;
    return;
  }
  if (g_out_fd >= 0 && g_out_wav && g_out_fd != 1 && g_out_bytes <= 0xfffffff0u && lseek(g_out_fd, 0, SEEK_SET) == 0) {
    unsigned char h[44];
    wav_header(h, g_out_rate, g_out_ch, 0, (uint32_t)g_out_bytes);
    write_all(g_out_fd, h, 44);
  }
}

/* one file through the reference's own loop (P:2566-2583): stdin, and files the whole-stream path declines */
static void cli_stream_file(pdmp3_handle* id, const char* filename, FILE* fp) {
  unsigned char out[INBUF_SIZE];
  pdmp3_open_feed(id);
  size_t done;
  int res;
  while ((res = pdmp3_read(id, out, INBUF_SIZE, &done)) != PDMP3_ERR) {
    write_raw(filename, out, done, done ? (long)kLsfSampleRates[sfreq9(&id->l_hdr)] : 0, id->l_hdr.mode == 3 ? 1 : 2);
    if (res == PDMP3_NEED_MORE) {
      unsigned char in[4096];
      const size_t n = fread(in, 1, sizeof in, fp);
      if (!n) break;
      (void)pdmp3_feed(id, in, n);
    }
  }
}

/* The streaming API driven from a memory buffer, in C (include/pdmp3_bulk.h): pdmp3_new, pdmp3_open_feed, then
 * pdmp3_read(read_bytes) until PDMP3_ERR, feeding feed_bytes on PDMP3_NEED_MORE -- the reference driver's loop
 * (P:2564-2584) with its two sizes as parameters (4096 / 16384 there).  eager != 0: the caller keeps the ring as full
 * as feed_bytes-sized feeds allow instead of waiting for PDMP3_NEED_MORE. */
long long pdmp3_amd_stream_loop(const unsigned char* mp3, size_t n, unsigned char* out, size_t cap,
                                size_t feed_bytes, size_t read_bytes, int eager) {
  if (!mp3 || !feed_bytes || !read_bytes) return -1;
  pdmp3_handle* id = pdmp3_new(NULL, NULL);
  if (!id) return -1;
  unsigned char* buf = (unsigned char*)malloc(read_bytes);
  if (!buf) { pdmp3_delete(id); return -1; }
  pdmp3_open_feed(id);
  size_t fed = 0, done, total = 0;
  int res;
  for (;;) {
    if (eager)
      while (fed < n) {
        const size_t take = n - fed < feed_bytes ? n - fed : feed_bytes;
        /* never to the last byte: a ring filled exactly looks EMPTY to the reference (iend == istart, P:1062-1068)
         * and the next feeds would overwrite it -- a caller of the real API has to keep count for this itself */
        if (take >= ring_free_logical(id) || pdmp3_feed(id, mp3 + fed, take) != PDMP3_OK) break;
        fed += take;
      }
    res = pdmp3_read(id, buf, read_bytes, &done);
    if (res == PDMP3_ERR) break;
    if (out && total < cap) memcpy(out + total, buf, done < cap - total ? done : cap - total);
    total += done;
    if (res == PDMP3_NEED_MORE) {
      if (fed >= n) break;
      if (!eager) {
        const size_t take = n - fed < feed_bytes ? n - fed : feed_bytes;
        (void)pdmp3_feed(id, mp3 + fed, take);
        fed += take;
      }
    }
  }
  free(buf);
  pdmp3_delete(id);
  return (long long)total;
}

/* Same contract as the reference's driver: every named file is decoded to interleaved int16 and appended to
 * "<first name>.raw".  Regular files take the whole-stream path (include/pdmp3_bulk.h), whose output is by
 * definition what the loop above produces -- parse state carried from file to file like the reference's single
 * handle does; PDMP3_CLI_STREAMING=1 forces the loop. */
void pdmp3(char* const* mp3s) {
  if (*mp3s && !strncmp("/dev/dsp", *mp3s, 8)) mp3s++;      /* OSS device argument accepted, playback not supported */
  pdmp3_handle* id = pdmp3_new(NULL, NULL);
  if (!id) { fputs("Cannot open stream API (no transform engine)\n", stderr); exit(0); }
  const char* force = getenv("PDMP3_CLI_STREAMING");
  const int streaming_only = force && *force && *force != '0';
  /* $PDMP3_CLI_ISO = mask of PDMP3_ISO_* (include/pdmp3.h): the standard's behaviour instead of the reference's; default 0 */
  const char* iso_env = getenv("PDMP3_CLI_ISO");
  const unsigned iso = iso_env ? (unsigned)strtoul(iso_env, NULL, 0) & (PDMP3_ISO_ALL | PDMP3_ISO_LSF) : 0u;
  (void)pdmp3_amd_set_quirks(id, iso);
  struct bulk* b = NULL;
  int bulk_used = 0, loop_used = 0;
  for (; *mp3s; mp3s++) {
    const char* filename = *mp3s;
    FILE* fp = strcmp(filename, "-") ? fopen(filename, "r") : stdin;
    if (!fp) { fputs("Cannot open file\n", stderr); exit(0); }
    unsigned char* data = NULL;
    long size = -1;
    /* the two paths keep their parse state in different handles: once one of them has decoded a file, later
     * files stay on it */
    if (fp != stdin && !streaming_only && !loop_used && fseek(fp, 0, SEEK_END) == 0 && (size = ftell(fp)) >= 0 &&
        fseek(fp, 0, SEEK_SET) == 0) {
      data = (unsigned char*)malloc((size_t)size + 1);
      if (data && fread(data, 1, (size_t)size, fp) != (size_t)size) { free(data); data = NULL; }
      if (!data) (void)fseek(fp, 0, SEEK_SET);
    }
    long long total = -1;
    if (data) {
      total = pdmp3_amd_scan_buffer_iso(data, (size_t)size, iso, NULL);
      if (total == PDMP3_BULK_REPLAY && bulk_used) {            /* the reference would not terminate on this file */
        fprintf(stderr, "pdmp3: %s: the reference decoder replays its input ring on this stream; skipped\n", filename);
        total = 0;
      }
    }
    if (data && total >= 0) {
      if (!b) { b = pdmp3_amd_bulk_new(0, 0); if (b) (void)pdmp3_amd_bulk_set_quirks(b, iso); }
      unsigned char* pcm = (unsigned char*)malloc((size_t)total + 1);
      if (!b || !pcm) { fputs("Cannot open stream API (no transform engine)\n", stderr); exit(0); }
      b->carry = bulk_used;                       /* first file: fresh state, like the reference's new handle */
      long rate = 0; int ch = 0;
      const long long got = pdmp3_amd_bulk_decode(b, data, (size_t)size, pcm, (size_t)total, &rate, &ch);
      if (got != total) { fputs("pdmp3: engine failure\n", stderr); exit(-1); }
      write_raw(filename, pcm, (size_t)total, rate, ch);
      free(pcm);
      bulk_used = 1;
    } else {
      if (data) (void)fseek(fp, 0, SEEK_SET);
      cli_stream_file(id, filename, fp);
      loop_used = 1;
    }
    free(data);
    if (fp != stdin) fclose(fp);
  }
  finish_output();
  if (b) pdmp3_amd_bulk_delete(b);
  pdmp3_delete(id);
}

