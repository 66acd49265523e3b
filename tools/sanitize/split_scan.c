/* sanitizer driver: the split scan's pre-pass, scanners and stitch order on one stream, K = 1 .. 8 (tools/sanitize/run.sh) */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
long long pdmp3_amd_test_split_scan(const unsigned char* mp3, size_t n, int window_frames, int K, unsigned iso,
                                    unsigned char* out, size_t out_cap, long long* frames);
int main(int argc, char** argv) {
  if (argc < 2) return 2;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 2;
  fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
  unsigned char* d = malloc((size_t)n + 1);
  if (fread(d, 1, (size_t)n, f) != (size_t)n) return 2;
  fclose(f);
  size_t cap = (size_t)n * 3 + (1u << 20);
  unsigned char* a = malloc(cap); unsigned char* b = malloc(cap);
  long long fa = 0, fb = 0;
  const long long na = pdmp3_amd_test_split_scan(d, (size_t)n, 64, 1, 0, a, cap, &fa);
  int bad = 0;
  for (int k = 2; k <= 8; k += 2) {
    char parts[8];
    snprintf(parts, sizeof parts, "%d", k == 2 ? 1 : k);                /* the pre-pass in 1, 4, 6, 8 parts (hop threads) */
    setenv("PDMP3_BULK_PREPASS_THREADS", parts, 1);
    const long long nb = pdmp3_amd_test_split_scan(d, (size_t)n, 64, k, 0, b, cap, &fb);
    const int same = na == nb && fa == fb && (na <= 0 || !memcmp(a, b, (size_t)na));
    printf("K %d, pre-pass in %s: %lld bytes, %lld frames, %s\n", k, parts, nb, fb, same ? "same as one scanner" : "DIFFERENT");
    bad += !same;
  }
  return bad;
}
