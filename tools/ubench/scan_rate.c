/* Host scan rate of the whole-stream decoder (stage A alone, compact pool form): ns per frame on this machine, for a
 * stream that does not fit the caches and for a cache-resident prefix of it.  Development tool.
 *   gcc -O2 -Iinclude tools/ubench/scan_rate.c -o /tmp/scan_rate -Lpdmp3_amd -lpdmp3 -lpdmp3_hip -Wl,-rpath,$PWD/pdmp3_amd
 *   /tmp/scan_rate file.mp3 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include "pdmp3_hip.h"
#include "pdmp3_bulk.h"
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
int main(int argc, char** argv) {
  if (argc < 2) return 2;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 2;
  fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
  unsigned char* m = malloc((size_t)n);
  if (fread(m, 1, (size_t)n, f) != (size_t)n) return 2;
  fclose(f);
  const size_t cap = (size_t)n / 96 + 16;
  pdmp3_frame_bits* bits = calloc(cap, sizeof *bits);
  pdmp3_row_desc* desc = calloc(cap, sizeof *desc);
  const size_t pc = cap * 2064 + 16384;
  uint8_t* pool = calloc(pc, 1);
  pdmp3_amd_bulk* b = pdmp3_amd_bulk_new_parse_bits();
  for (long part = n; part > 200000; part /= 8) {
    double best = 1e9; long long fr = 0; size_t used = 0;
    for (int r = 0; r < 12; r++) {
      const double t = now();
      fr = pdmp3_amd_bulk_parse_pool(b, m, (size_t)part, bits, desc, pool, pc, cap, &used);
      const double d = now() - t;
      if (d < best) best = d;
    }
    printf("%ld bytes: %lld frames, %.1f ns/frame, pool %zu bytes\n", part, fr, best / (double)fr * 1e9, used);
  }
  return 0;
}
