/* Several handles reading at once through pdmp3_feed / pdmp3_read (parse-only handles: the host stage alone, no GPU):
 * frames per second of each, so that how the library's helper threads are shared shows (include/pdmp3.h).
 *   gcc -O2 -Iinclude tools/ubench/stream_share.c -o /tmp/stream_share -Lpdmp3_amd -lpdmp3 -lpdmp3_hip -lpthread -Wl,-rpath,$PWD/pdmp3_amd
 *   /tmp/stream_share file.mp3 [handles = 2] [read bytes = 65536] */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "pdmp3.h"
pdmp3_handle* pdmp3_amd_new_parse_only(void);
static unsigned char* mp3; static size_t n, read_bytes = 65536;
static pthread_barrier_t bar;
typedef struct { double secs; size_t pcm; } result;
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }
static void* run(void* arg) {
  result* r = (result*)arg;
  pdmp3_handle* id = pdmp3_amd_new_parse_only(); pdmp3_open_feed(id);
  unsigned char* buf = malloc(read_bytes); size_t fed = 0, done, total = 0; int res;
  pthread_barrier_wait(&bar);
  const double t0 = now_s();
  for (;;) {
    res = pdmp3_read(id, buf, read_bytes, &done); if (res == PDMP3_ERR) break; total += done;
    if (res == PDMP3_NEED_MORE) { if (fed >= n) break; size_t take = n - fed < 4096 ? n - fed : 4096; while (take && pdmp3_feed(id, mp3 + fed, take) == PDMP3_OK) { fed += take; take = n - fed < 4096 ? n - fed : 4096; } }
  }
  r->secs = now_s() - t0; r->pcm = total;
  pdmp3_delete(id); free(buf); return NULL;
}
int main(int argc, char** argv) {
  if (argc < 2) return 2;
  FILE* f = fopen(argv[1], "rb"); if (!f) return 2;
  fseek(f, 0, SEEK_END); n = ftell(f); fseek(f, 0, SEEK_SET); mp3 = malloc(n); if (fread(mp3, 1, n, f) != n) return 1;
  const int h = argc > 2 ? atoi(argv[2]) : 2;
  if (argc > 3) read_bytes = (size_t)atol(argv[3]);
  pthread_t th[16]; result res[16];
  if (h < 1 || h > 16) return 2;
  pthread_barrier_init(&bar, NULL, h);
  for (int i = 0; i < h; i++) pthread_create(&th[i], NULL, run, &res[i]);
  for (int i = 0; i < h; i++) pthread_join(th[i], NULL);
  double sum = 0;
  printf("{\"handles\": %d, \"read_bytes\": %zu, \"k_frames_per_s\": [", h, read_bytes);
  for (int i = 0; i < h; i++) { const double r = res[i].pcm / 4608.0 / res[i].secs * 1e-3; sum += r; printf("%s%.1f", i ? ", " : "", r); }
  printf("], \"sum\": %.1f}\n", sum);
  return 0;
}
