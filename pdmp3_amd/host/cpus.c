/* cpus.c -- libpdmp3.so: the CPUs this process can keep busy, and those next to the GPU (thread counts and binding of the
 * whole-stream decoder's threads).  See host_internal.h for the map of the library. */
#include "bulk_internal.h"


/* CPUs this process can keep busy: the affinity mask, capped by the cgroup CPU quota (containers often show every
 * CPU of the host but run under a quota of a few; threads beyond it only get throttled) */
int usable_cpus(void) {
  long n = sysconf(_SC_NPROCESSORS_ONLN);
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0 && CPU_COUNT(&set) < n) n = CPU_COUNT(&set);
  FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r");
  if (f) {
    char q[32];
    long long period = 0;
    if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") && period > 0) {
      const long long quota = atoll(q);
      const long c = (long)((quota + period / 2) / period);
      if (c >= 1 && c < n) n = c;
    }
    fclose(f);
  }
  return n < 1 ? 1 : (int)n;
}

/* The CPUs next to the GPU (/sys/bus/pci/devices/<address>/local_cpulist), as far as this process may run on them.
 * The GPU boxes are two-socket machines: the decoder's pinned buffers sit behind one socket's memory controllers and
 * the GPU behind one socket's PCIe root; helper threads that wander to the other socket copy every PCM byte across the
 * socket link and back (measured on such a box, PCM to pageable memory, the same build run after run: 6.1 .. 9.1 M
 * frames/s; the round-4 driver run's 6.4 M against 9.7 M elsewhere).  So the decoder's own threads -- copy pool,
 * submitter, gather helpers, the split scan's crew -- are kept on the GPU's node, and its pinned buffers are allocated
 * from a thread that is there.  PDMP3_BULK_NUMA=0: leave everything to the scheduler.  Returns the number of CPUs. */
int gpu_local_cpus(pdmp3_hip_ctx* ctx, cpu_set_t* out) {
  CPU_ZERO(out);
  const char* e = getenv("PDMP3_BULK_NUMA");
  if (e && *e == '0') return 0;
  char bdf[64], path[160], list[1024];
  if (pdmp3_hip_pci_bus_id(ctx, bdf, (int)sizeof bdf) != PDMP3_HIP_OK || !bdf[0]) return 0;
  for (char* p = bdf; *p; p++) if (*p >= 'A' && *p <= 'F') *p = (char)(*p - 'A' + 'a');
  snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/local_cpulist", bdf);
  FILE* f = fopen(path, "r");
  if (!f) return 0;
  const int ok = fgets(list, sizeof list, f) != NULL;
  fclose(f);
  if (!ok) return 0;
  cpu_set_t mine;
  if (sched_getaffinity(0, sizeof mine, &mine) != 0) return 0;
  for (const char* p = list; *p;) {                     /* "0-63,128-191" */
    char* end;
    const long a = strtol(p, &end, 10);
    if (end == p) break;
    long b = a;
    p = end;
    if (*p == '-') { b = strtol(p + 1, &end, 10); p = end; }
    for (long c = a; c <= b && c < CPU_SETSIZE; c++) if (c >= 0 && CPU_ISSET((int)c, &mine)) CPU_SET((int)c, out);
    while (*p == ',' || *p == ' ' || *p == '\n') p++;
  }
  const int n = CPU_COUNT(out);
  if (n == CPU_COUNT(&mine)) { CPU_ZERO(out); return 0; }           /* one node, or already bound: nothing to do */
  return n;
}
void bind_thread(pthread_t t, const cpu_set_t* set) { if (CPU_COUNT(set) > 0) (void)pthread_setaffinity_np(t, sizeof *set, set); }
