#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include "pdmp3.h"
pdmp3_handle* pdmp3_amd_new_parse_only(void);
static unsigned char* mp3; static size_t n;
static void* run(void* arg) {
  size_t read_bytes = (size_t)arg;
  pdmp3_handle* id = pdmp3_amd_new_parse_only(); pdmp3_open_feed(id);
  if (getenv("PDMP3_SAN_ISO")) pdmp3_amd_set_quirks(id, (unsigned)strtoul(getenv("PDMP3_SAN_ISO"), 0, 0));   /* e.g. 0x7f: the ISO switches + LSF */
  unsigned char* buf = malloc(read_bytes); size_t fed = 0, done, total = 0; int res;
  for (;;) {
    res = pdmp3_read(id, buf, read_bytes, &done); if (res == PDMP3_ERR) break; total += done;
    if (res == PDMP3_NEED_MORE) { if (fed >= n) break; for (int k = 0; k < 7 && fed < n; k++) { size_t take = n - fed < 2048 ? n - fed : 2048; if (pdmp3_feed(id, mp3 + fed, take) != PDMP3_OK) break; fed += take; } }
  }
  printf("total %zu\n", total); pdmp3_delete(id); free(buf); return NULL;
}
int main(int argc, char** argv) {
  FILE* f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); n = ftell(f); fseek(f, 0, SEEK_SET); mp3 = malloc(n); if (fread(mp3, 1, n, f) != n) return 1;
  n = n > 3000000 ? 3000000 : n;
  pthread_t th[3]; size_t rb[3] = {65536, 16384, 40000};
  for (int i = 0; i < 3; i++) pthread_create(&th[i], NULL, run, (void*)rb[i]);
  for (int i = 0; i < 3; i++) pthread_join(th[i], NULL);
  return 0;
}
