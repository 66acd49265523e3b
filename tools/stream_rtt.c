/* development: round trip of one streaming batch (pdmp3_hip_stream_submit + pdmp3_hip_stream_wait) for n frames,
 * no parsing -- what the synchronous drop-in API pays per read-ahead batch besides its own host work.
 * build + run on the GPU box:  gcc -O2 -Iinclude -o /tmp/rtt tools/stream_rtt.c -Lpdmp3_amd -lpdmp3_hip -Wl,-rpath,$PWD/pdmp3_amd && /tmp/rtt */
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include "pdmp3_hip.h"
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
int main(void) {
  pdmp3_hip_ctx* ctx; pdmp3_hip_stream* hs;
  if (pdmp3_hip_create(0, &ctx) != PDMP3_HIP_OK || pdmp3_hip_stream_create(ctx, 16, &hs) != PDMP3_HIP_OK) { fprintf(stderr, "%s\n", pdmp3_hip_last_error()); return 1; }
  pdmp3_host_generate_frames(0x5EED, 0, 16, pdmp3_hip_stream_spectra(hs), pdmp3_hip_stream_side(hs));
  const int ns[] = {1, 2, 4, 8, 14, 16};
  for (unsigned k = 0; k < sizeof ns / sizeof *ns; k++) {
    const int n = ns[k], reps = 3000;
    double best = 1e9, sub = 0, wt = 0;
    for (int r = 0; r < reps + 100; r++) {
      const double t0 = now();
      if (pdmp3_hip_stream_submit(hs, 0, n) != PDMP3_HIP_OK) { fprintf(stderr, "%s\n", pdmp3_hip_last_error()); return 1; }
      const double t1 = now();
      pdmp3_hip_stream_wait(hs, 0);
      const double t2 = now();
      if (r >= 100) { sub += t1 - t0; wt += t2 - t1; if (t2 - t0 < best) best = t2 - t0; }
    }
    printf("n %2d  submit %.2f us  wait %.2f us  total mean %.2f us  best %.2f us\n", n, sub / reps * 1e6, wt / reps * 1e6, (sub + wt) / reps * 1e6, best * 1e6);
  }
  pdmp3_hip_stream_destroy(hs); pdmp3_hip_destroy(ctx);
  return 0;
}
