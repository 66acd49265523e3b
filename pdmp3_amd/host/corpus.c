/* corpus.c -- libpdmp3.so: a corpus of files over the GPUs of a node (include/pdmp3_bulk.h pdmp3_amd_corpus_*).
 * See host_internal.h for the map of the library. */
#include "bulk_internal.h"

/* ------------------------------------------------------------------------ */
/* A corpus of files over the GPUs of a node ("C4: whole files per */
/* GPU, largest first"): the C form of pdmp3_amd/sharding.py assign_files +   */
/* one whole-stream decoder per device.  include/pdmp3_bulk.h                  */
/* ------------------------------------------------------------------------ */
/* largest-first greedy; ties go to the lower rank (== sharding.assign_files): rank_of[i] = the device slot of file i */
void pdmp3_amd_corpus_assign(const size_t* sizes, int n_files, int world, int* rank_of) {
  if (!sizes || !rank_of || n_files <= 0 || world <= 0) return;
  int* order = (int*)malloc((size_t)n_files * sizeof *order);
  unsigned long long* load = (unsigned long long*)calloc((size_t)world, sizeof *load);
  if (!order || !load) { free(order); free(load); for (int i = 0; i < n_files; i++) rank_of[i] = i % world; return; }
  for (int i = 0; i < n_files; i++) order[i] = i;
  for (int i = 1; i < n_files; i++) {             /* stable insertion sort by size, descending (corpora are thousands of files at most) */
    const int k = order[i];
    int j = i;
    while (j > 0 && sizes[order[j - 1]] < sizes[k]) { order[j] = order[j - 1]; j--; }
    order[j] = k;
  }
  for (int q = 0; q < n_files; q++) {
    int r = 0;
    for (int k = 1; k < world; k++) if (load[k] < load[r]) r = k;
    rank_of[order[q]] = r;
    load[r] += sizes[order[q]];
  }
  free(order); free(load);
}

struct corpus_job {
  int device, slot, world, n_files, host_huffman, threads, window;
  unsigned iso;
  const int* rank_of;
  const unsigned char* const* mp3s; const size_t* sizes;
  unsigned char* const* pcm; const size_t* pcm_caps; long long* pcm_bytes;
  int rc;
};
static void* corpus_worker(void* arg) {
  struct corpus_job* j = (struct corpus_job*)arg;
  struct bulk* b = pdmp3_amd_bulk_new_on(j->threads, j->window, j->host_huffman, j->device);
  if (!b) { j->rc = -1; return NULL; }
  (void)pdmp3_amd_bulk_set_quirks(b, j->iso);
  for (int i = 0; i < j->n_files && j->rc == 0; i++) {
    if (j->rank_of[i] != j->slot) continue;
    /* (asynchronous: the next file's scan runs under this one's GPU work and copy-out) */
    const long long got = pdmp3_amd_bulk_decode_async(b, j->mp3s[i], j->sizes[i], j->pcm[i], j->pcm_caps[i], NULL, NULL);
    j->pcm_bytes[i] = got;
    if (got < 0 && got != PDMP3_BULK_REPLAY) j->rc = -1;
  }
  if (pdmp3_amd_bulk_wait(b) != 0) j->rc = -1;
  pdmp3_amd_bulk_delete(b);
  return NULL;
}
/* n_files whole streams over n_devices HIP devices (a device may be listed more than once: that many decoders share it), one
 * host thread and one whole-stream decoder per entry, files dealt largest first.  pcm[i] (capacity pcm_caps[i]) receives file
 * i's PCM, pcm_bytes[i] what pdmp3() would write for it (or PDMP3_BULK_REPLAY).  0, or -1 on an engine failure. */
int pdmp3_amd_corpus_decode(const int* devices, int n_devices, const unsigned char* const* mp3s, const size_t* sizes, int n_files,
                            unsigned char* const* pcm, const size_t* pcm_caps, long long* pcm_bytes, unsigned iso_mask,
                            int threads_per_decoder, int window_frames, int host_huffman) {
  if (!devices || n_devices < 1 || n_devices > 64 || n_files < 0 || (n_files && (!mp3s || !sizes || !pcm || !pcm_caps || !pcm_bytes))) return -1;
  if (!n_files) return 0;
  int* rank_of = (int*)malloc((size_t)n_files * sizeof *rank_of);
  if (!rank_of) return -1;
  pdmp3_amd_corpus_assign(sizes, n_files, n_devices, rank_of);
  struct corpus_job jobs[64];
  pthread_t th[64];
  int started = 0, rc = 0;
  for (int k = 0; k < n_devices; k++) {
    jobs[k] = (struct corpus_job){devices[k], k, n_devices, n_files, host_huffman, threads_per_decoder, window_frames, iso_mask,
                                  rank_of, mp3s, sizes, pcm, pcm_caps, pcm_bytes, 0};
    if (pthread_create(&th[k], NULL, corpus_worker, &jobs[k]) != 0) { rc = -1; break; }
    started++;
  }
  for (int k = 0; k < started; k++) { pthread_join(th[k], NULL); if (jobs[k].rc) rc = -1; }
  free(rank_of);
  return rc;
}

