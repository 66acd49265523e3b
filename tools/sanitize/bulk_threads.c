#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "pdmp3_bulk.h"
int main(int argc, char** argv) {
  FILE* f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); size_t n = ftell(f); fseek(f, 0, SEEK_SET); unsigned char* mp3 = malloc(n); if (fread(mp3, 1, n, f) != n) return 1;
  if (n > 6000000) n = 6000000;
  size_t cap = 8000;
  int16_t* sp = malloc(cap * 2304 * 2); pdmp3_gc_side* sd = malloc(cap * 4 * sizeof *sd);
  for (int th = 2; th <= 6; th += 2) {
    pdmp3_amd_bulk* b = pdmp3_amd_bulk_new_parse_only(th, 64);
    if (getenv("PDMP3_SAN_ISO")) pdmp3_amd_bulk_set_quirks(b, (unsigned)strtoul(getenv("PDMP3_SAN_ISO"), 0, 0));   /* e.g. 0x7f: the ISO switches + LSF */
    long long pcm = 0;
    for (int rep = 0; rep < 2; rep++) { long long fr = pdmp3_amd_bulk_parse(b, mp3, n, sp, sd, cap, &pcm); printf("threads %d frames %lld\n", th, fr); }
    pdmp3_amd_bulk_delete(b);
  }
  return 0;
}
