// node.hip -- include/pdmp3_node.h: one stream decoded by the GPUs of one node.
//
// One host thread per rank drives its device through the engine's own C-ABI (pdmp3_hip_create / _generate_frames /
// _decode_frames: nothing here reaches into the engine); the shards' PCM is gathered to rank 0's device by grouped
// ncclSend / ncclRecv (RCCL, looked up with dlopen: the library does not link against it) or, for tests that run several
// ranks on one GPU, by device-to-device copies.  No reference counterpart (pdmp3.c has no parallelism).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>
#include <chrono>
#include <thread>
#include <vector>

#include <rccl/rccl.h>                    // types and prototypes only: every entry point is taken from dlsym
#include "../../include/pdmp3_node.h"

extern "C" void pdmp3_hip_set_error_(const char* text);

namespace {

struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

// the copy of RCCL the process has already (a Python process that imported torch has torch's), else the system's
bool rccl_load(Rccl& r) {
  static const char* names[] = {"librccl.so.1", "librccl.so"};
  for (int pass = 0; pass < 2 && !r.lib; pass++)
    for (const char* n : names) {
      r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
      if (r.lib) break;
    }
  if (!r.lib) return false;
#define PD_SYM(field, name) *(void**)(&r.field) = dlsym(r.lib, name); if (!r.field) return false;
  PD_SYM(CommInitAll, "ncclCommInitAll") PD_SYM(CommDestroy, "ncclCommDestroy") PD_SYM(CommCount, "ncclCommCount")
  PD_SYM(GroupStart, "ncclGroupStart") PD_SYM(GroupEnd, "ncclGroupEnd") PD_SYM(Send, "ncclSend") PD_SYM(Recv, "ncclRecv")
  PD_SYM(GetErrorString, "ncclGetErrorString")
#undef PD_SYM
  return true;
}

struct Rank {
  int device = 0;
  pdmp3_hip_ctx* ctx = nullptr;
  hipStream_t stream = nullptr;      // the rank's kernels
  hipStream_t xfer = nullptr;        // its share of the exchange: a slice's PCM leaves while the next slice decodes
  ncclComm_t comm = nullptr;
  void* d_state = nullptr;           // synthesis state carried from slice to slice of the shard (pdmp3_hip_state_bytes)
  hipEvent_t ev_slice = nullptr;     // "this slice's PCM is complete" (recorded on stream, waited for on xfer)
  hipEvent_t ev_d0 = nullptr, ev_d1 = nullptr, ev_x0 = nullptr, ev_x1 = nullptr;   // timing: decode / exchange, first start to last end
  long long moved = 0;               // bytes this rank sent to another device in this call
  // the shard's buffers on the device, kept from call to call (grown when a larger shard comes)
  int16_t* d_spectra = nullptr;
  pdmp3_gc_side* d_side = nullptr;
  int16_t* d_pcm = nullptr;
  long long cap = 0;
  // this call
  long long first = 0, count = 0, discard = 0, lo = 0;
  int rc = PDMP3_HIP_OK;
  char err[256] = "";
};

// every pdmp3_node_* entry point runs hipSetDevice for its ranks on the CALLER's thread: the caller's current device is
// put back on every way out (a Python caller's torch.cuda.current_device() moved to the last GPU)
struct DeviceGuard {
  int dev = -1;
  DeviceGuard() { if (hipGetDevice(&dev) != hipSuccess) dev = -1; }
  ~DeviceGuard() { if (dev >= 0) (void)hipSetDevice(dev); }
};

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int node_fail(int code, const char* fmt, const char* a = "", const char* b = "") {
  char t[256];
  snprintf(t, sizeof t, fmt, a, b);
  pdmp3_hip_set_error_(t);
  return code;
}

}  // namespace

struct pdmp3_node {
  int n = 0, transport = PDMP3_NODE_RCCL;
  int slices = 8;                    // $PDMP3_NODE_SLICES: pieces a shard is decoded and sent in (include/pdmp3_node.h)
  std::vector<Rank> rank;
  Rccl rccl;
  int rccl_ranks = 0;
};

extern "C" int pdmp3_node_ranks(const pdmp3_node* node) { return node ? node->n : 0; }

// == pdmp3_amd/sharding.py frame_range / halo_start / shard_with_halo
extern "C" void pdmp3_node_shard(long long n_frames, int rank, int world, const uint8_t* flags,
                                 long long* first_out, long long* count_out, long long* discard_out) {
  const long long base = n_frames / world, rem = n_frames % world;
  const long long lo = rank * base + (rank < rem ? rank : rem);
  const long long hi = lo + base + (rank < rem ? 1 : 0);
  long long first = lo - 2 > 0 ? lo - 2 : 0;                          // the fixed halo: two frames
  if (flags && lo > 0) {
    auto mono = [&](long long f) { return ((flags[f] & PDMP3_FR_MODE_MASK) >> PDMP3_FR_MODE_SHIFT) == 3; };
    auto reset = [&](long long f) { return (flags[f] & PDMP3_FR_RESET) != 0; };
    if (mono(lo - 1) && !reset(lo - 1)) {
      // the frame before the cut is mono: channel 1's state is what the last stereo frame left, however far back
      // (the reference's store[ch] / v_vec[ch] are per channel, pdmp3.c:1777, 2126); the kernel's pre-halo wants to see
      // the frame in front of that one too (its H5 corner), unless that frame starts from zero anyway
      long long f = lo - 2;
      while (f >= 0 && mono(f) && !reset(f)) f--;
      if (f >= 0 && !mono(f)) {
        const long long back = (reset(f) || f == 0) ? f : f - 1;
        if (back < first) first = back;
      }
    }
  }
  if (first_out) *first_out = first;
  if (count_out) *count_out = hi - first;
  if (discard_out) *discard_out = lo - first;
}

extern "C" void pdmp3_node_destroy(pdmp3_node* node) {
  if (!node) return;
  DeviceGuard guard;
  for (Rank& r : node->rank) {
    (void)hipSetDevice(r.device);
    if (r.stream) (void)hipStreamSynchronize(r.stream);
    if (r.xfer) (void)hipStreamSynchronize(r.xfer);
    if (r.comm && node->rccl.CommDestroy) (void)node->rccl.CommDestroy(r.comm);
    (void)hipFree(r.d_spectra); (void)hipFree(r.d_side); (void)hipFree(r.d_pcm); (void)hipFree(r.d_state);
    for (hipEvent_t e : {r.ev_slice, r.ev_d0, r.ev_d1, r.ev_x0, r.ev_x1}) if (e) (void)hipEventDestroy(e);
    if (r.xfer) (void)hipStreamDestroy(r.xfer);
    if (r.stream) (void)hipStreamDestroy(r.stream);
    if (r.ctx) pdmp3_hip_destroy(r.ctx);
  }
  delete node;
}

extern "C" int pdmp3_node_create(const int* devices, int n, int transport, pdmp3_node** out) {
  if (!devices || n < 1 || n > 64 || !out || (transport != PDMP3_NODE_RCCL && transport != PDMP3_NODE_COPY))
    return node_fail(PDMP3_HIP_EINVAL, "pdmp3_node_create: bad argument");
  *out = nullptr;
  if (transport == PDMP3_NODE_RCCL)
    for (int i = 0; i < n; i++)
      for (int j = 0; j < i; j++)
        if (devices[i] == devices[j])
          return node_fail(PDMP3_HIP_EINVAL, "pdmp3_node_create: a device listed twice needs PDMP3_NODE_COPY (RCCL takes every GPU once)");
  DeviceGuard guard;
  pdmp3_node* node = new pdmp3_node;
  node->n = n; node->transport = transport;
  if (const char* e = getenv("PDMP3_NODE_SLICES")) { const int v = atoi(e); if (v >= 1 && v <= 64) node->slices = v; }
  node->rank.resize((size_t)n);
  for (int i = 0; i < n; i++) {
    Rank& r = node->rank[(size_t)i];
    r.device = devices[i];
    if (pdmp3_hip_create(r.device, &r.ctx) != PDMP3_HIP_OK) { pdmp3_node_destroy(node); return PDMP3_HIP_EDEVICE; }   // (text: the engine's)
    if (hipSetDevice(r.device) != hipSuccess || hipStreamCreateWithFlags(&r.stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&r.xfer, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&r.ev_slice, hipEventDisableTiming) != hipSuccess ||
        hipEventCreate(&r.ev_d0) != hipSuccess || hipEventCreate(&r.ev_d1) != hipSuccess ||
        hipEventCreate(&r.ev_x0) != hipSuccess || hipEventCreate(&r.ev_x1) != hipSuccess ||
        hipMalloc(&r.d_state, pdmp3_hip_state_bytes()) != hipSuccess) {
      pdmp3_node_destroy(node);
      return node_fail(PDMP3_HIP_EDEVICE, "pdmp3_node_create: streams / events / state block");
    }
  }
  if (transport == PDMP3_NODE_RCCL) {
    if (!rccl_load(node->rccl)) {
      pdmp3_node_destroy(node);
      return node_fail(PDMP3_HIP_EDEVICE, "pdmp3_node_create: no RCCL library (librccl.so.1 / librccl.so): %s", dlerror() ? dlerror() : "symbols missing");
    }
    std::vector<ncclComm_t> comms((size_t)n);
    const ncclResult_t rc = node->rccl.CommInitAll(comms.data(), n, devices);
    if (rc != ncclSuccess) {
      const char* what = node->rccl.GetErrorString(rc);
      pdmp3_node_destroy(node);
      return node_fail(PDMP3_HIP_EDEVICE, "pdmp3_node_create: ncclCommInitAll: %s", what);
    }
    for (int i = 0; i < n; i++) node->rank[(size_t)i].comm = comms[(size_t)i];
    int cnt = 0;
    if (node->rccl.CommCount(comms[0], &cnt) == ncclSuccess) node->rccl_ranks = cnt;
  }
  *out = node;
  return PDMP3_HIP_OK;
}

namespace {

int rank_reserve(Rank& r, long long frames) {
  if (frames <= r.cap) return PDMP3_HIP_OK;
  (void)hipFree(r.d_spectra); (void)hipFree(r.d_side); (void)hipFree(r.d_pcm);
  r.d_spectra = nullptr; r.d_side = nullptr; r.d_pcm = nullptr; r.cap = 0;
  if (hipMalloc((void**)&r.d_spectra, (size_t)frames * PDMP3_FRAME_SPECTRA_BYTES) != hipSuccess ||
      hipMalloc((void**)&r.d_side, (size_t)frames * PDMP3_FRAME_SIDE_BYTES) != hipSuccess ||
      hipMalloc((void**)&r.d_pcm, (size_t)frames * PDMP3_FRAME_PCM_BYTES) != hipSuccess) {
    snprintf(r.err, sizeof r.err, "pdmp3_node: hipMalloc of a shard of %lld frames failed on device %d", frames, r.device);
    return PDMP3_HIP_ENOMEM;
  }
  r.cap = frames;
  return PDMP3_HIP_OK;
}

// every rank's `body` on a host thread of its own, device set; returns the first failure
template <class F>
int on_all_ranks(pdmp3_node* node, F body) {
  std::vector<std::thread> th;
  for (int k = 0; k < node->n; k++)
    th.emplace_back([node, k, &body] {
      Rank& r = node->rank[(size_t)k];
      r.rc = PDMP3_HIP_OK; r.err[0] = 0;
      if (hipSetDevice(r.device) != hipSuccess) { r.rc = PDMP3_HIP_EDEVICE; snprintf(r.err, sizeof r.err, "pdmp3_node: hipSetDevice(%d) failed", r.device); return; }
      body(k, r);
      if (r.rc != PDMP3_HIP_OK && !r.err[0]) snprintf(r.err, sizeof r.err, "%s", pdmp3_hip_last_error());   // (the engine's text is per thread: this one's)
    });
  for (std::thread& t : th) t.join();
  for (Rank& r : node->rank) if (r.rc != PDMP3_HIP_OK) { pdmp3_hip_set_error_(r.err); return r.rc; }
  return PDMP3_HIP_OK;
}

// Decode and exchange, pipelined (SURVEY 8e's one collective, BASELINE configs[4] "with RCCL PCM gather"): a rank's shard
// is decoded in `slices` consecutive pieces -- the synthesis state goes from piece to piece through the rank's state
// block, so the pieces cost no halo -- and piece i's PCM leaves on the rank's SECOND stream as soon as its kernel is done
// (an event), under the kernels of pieces i + 1 ...: at the C5 shape the exchange (7 x 576 MB into one GPU, ~3.8 ms over
// seven xGMI links by SURVEY 5's arithmetic) is 4 x the decode (0.93 ms per shard), so what the pipeline buys is the decode
// hidden under the exchange and the exchange started a slice's worth early, not the other way round.
//   RCCL:  sender k: ncclSend(piece) on comm k / stream xfer k;  root: ncclRecv(piece from k) for every k, one group per
//          piece, on the root's xfer stream -- point-to-point pairs, matched in order per pair.
//   COPY:  hipMemcpyPeerAsync on the sender's xfer stream (several ranks on one GPU: the tests' transport).
// Rank 0's own pieces are device-to-device copies on its xfer stream.  Every thread ends by draining its two streams.
int decode_and_gather(pdmp3_node* node, int16_t* d_pcm, pdmp3_node_timing* t) {
  const double t0 = now_ms();
  Rank& root = node->rank[0];
  // the same cut for every rank: piece j of a shard = frames [j q, (j + 1) q) of its [0, count), q from the LARGEST shard
  // (shards differ by at most one frame + their halos), so that the root's receives pair with the senders' sends
  long long longest = 0;
  for (Rank& r : node->rank) if (r.count > longest) longest = r.count;
  int S = node->slices;
  while (S > 1 && longest / S < 4096) S--;                       // no piece under 4096 frames: below that a launch does not fill the GPU
  const long long q = (longest + S - 1) / S;
  auto piece = [&](const Rank& r, int j, long long* a, long long* b) {   // frames [a, b) of the rank's buffer that piece j SENDS
    long long lo = (long long)j * q, hi = lo + q;
    if (hi > r.count) hi = r.count;
    if (lo < r.discard) lo = r.discard;                          // the halo's PCM stays behind
    *a = lo; *b = hi > lo ? hi : lo;
  };
  const int rc = on_all_ranks(node, [&](int k, Rank& r) {
    const Rccl& R = node->rccl;
    const bool rccl = node->transport == PDMP3_NODE_RCCL && node->n > 1;
    r.moved = 0;
    auto fail = [&](const char* what) { r.rc = PDMP3_HIP_EDEVICE; snprintf(r.err, sizeof r.err, "pdmp3_node: %s (device %d)", what, r.device); };
    if (hipMemsetAsync(r.d_state, 0, pdmp3_hip_state_bytes(), r.stream) != hipSuccess) return fail("hipMemsetAsync of the state block");
    if (hipEventRecord(r.ev_d0, r.stream) != hipSuccess || hipEventRecord(r.ev_x0, r.xfer) != hipSuccess) return fail("hipEventRecord");
    for (int j = 0; j < S; j++) {
      const long long lo = (long long)j * q, hi = lo + q < r.count ? lo + q : r.count;
      if (hi > lo) {
        r.rc = pdmp3_hip_decode_frames(r.ctx, r.d_spectra + (size_t)lo * 2304, r.d_side + (size_t)lo * 4, (int)(hi - lo), r.d_state,
                                       r.d_pcm + (size_t)lo * 2304, 0, r.stream);
        if (r.rc != PDMP3_HIP_OK) return;
      }
      if (hipEventRecord(r.ev_slice, r.stream) != hipSuccess || hipStreamWaitEvent(r.xfer, r.ev_slice, 0) != hipSuccess) return fail("slice event");
      long long a, b;
      piece(r, j, &a, &b);
      const size_t bytes = (size_t)(b - a) * PDMP3_FRAME_PCM_BYTES;
      const char* src = (const char*)r.d_pcm + (size_t)a * PDMP3_FRAME_PCM_BYTES;
      char* dst = (char*)d_pcm + (size_t)(r.first + a) * PDMP3_FRAME_PCM_BYTES;
      if (k == 0) {
        if (bytes && hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, r.xfer) != hipSuccess) return fail("copy of rank 0's PCM");
        if (rccl) {
          // the root's receives for piece j of every other rank: one group, side by side on its ingress links
          ncclResult_t g = R.GroupStart();
          for (int m = 1; m < node->n && g == ncclSuccess; m++) {
            const Rank& o = node->rank[(size_t)m];
            long long oa, ob;
            piece(o, j, &oa, &ob);
            if (ob > oa) g = R.Recv((char*)d_pcm + (size_t)(o.first + oa) * PDMP3_FRAME_PCM_BYTES, (size_t)(ob - oa) * PDMP3_FRAME_PCM_BYTES, ncclChar, m, r.comm, r.xfer);
          }
          const ncclResult_t g2 = R.GroupEnd();
          if (g != ncclSuccess || g2 != ncclSuccess) { r.rc = PDMP3_HIP_EDEVICE; snprintf(r.err, sizeof r.err, "pdmp3_node: ncclRecv: %s", R.GetErrorString(g != ncclSuccess ? g : g2)); return; }
        }
      } else if (bytes) {
        if (rccl) {
          const ncclResult_t g = R.Send(src, bytes, ncclChar, 0, r.comm, r.xfer);
          if (g != ncclSuccess) { r.rc = PDMP3_HIP_EDEVICE; snprintf(r.err, sizeof r.err, "pdmp3_node: ncclSend: %s", R.GetErrorString(g)); return; }
        } else if (hipMemcpyPeerAsync(dst, root.device, src, r.device, bytes, r.xfer) != hipSuccess) return fail("hipMemcpyPeerAsync");
        r.moved += (long long)bytes;
      }
    }
    if (hipEventRecord(r.ev_d1, r.stream) != hipSuccess || hipEventRecord(r.ev_x1, r.xfer) != hipSuccess) return fail("hipEventRecord");
    if (hipStreamSynchronize(r.stream) != hipSuccess || hipStreamSynchronize(r.xfer) != hipSuccess) return fail("decode / exchange failed");
  });
  if (rc != PDMP3_HIP_OK) return rc;
  if (t) {
    t->total_ms = now_ms() - t0;
    t->slices = S;
    t->rccl_ranks = node->rccl_ranks;
    for (Rank& r : node->rank) {
      float d = 0.0f, x = 0.0f;
      (void)hipSetDevice(r.device);
      if (hipEventElapsedTime(&d, r.ev_d0, r.ev_d1) == hipSuccess && d > t->decode_ms) t->decode_ms = d;
      if (hipEventElapsedTime(&x, r.ev_x0, r.ev_x1) == hipSuccess && x > t->gather_ms) t->gather_ms = x;
      t->gather_bytes += r.moved;
    }
  }
  return PDMP3_HIP_OK;
}

}  // namespace

extern "C" int pdmp3_node_decode_records(pdmp3_node* node, const int16_t* spectra, const pdmp3_gc_side* side, long long n_frames,
                                         int16_t* d_pcm, pdmp3_node_timing* t) {
  if (!node || n_frames < 0 || (n_frames && (!spectra || !side || !d_pcm))) return node_fail(PDMP3_HIP_EINVAL, "pdmp3_node_decode_records: bad argument");
  if (t) memset(t, 0, sizeof *t);
  if (!n_frames) return PDMP3_HIP_OK;
  DeviceGuard guard;
  if (n_frames / node->n + 8 > 0x7fffffffLL) return node_fail(PDMP3_HIP_EINVAL, "pdmp3_node_decode_records: shards of more than 2^31 frames");
  // the cut's way back past mono frames is read off the records' own flag bytes (any record of a frame carries them)
  std::vector<uint8_t> flags((size_t)n_frames);
  for (long long f = 0; f < n_frames; f++) flags[(size_t)f] = side[(size_t)f * 4].frame;
  for (int k = 0; k < node->n; k++) {
    Rank& r = node->rank[(size_t)k];
    pdmp3_node_shard(n_frames, k, node->n, flags.data(), &r.first, &r.count, &r.discard);
    r.lo = r.first + r.discard;
  }
  const double t0 = now_ms();
  int rc = on_all_ranks(node, [&](int, Rank& r) {
    if (r.count <= 0) return;
    if ((r.rc = rank_reserve(r, r.count)) != PDMP3_HIP_OK) return;
    // a mono frame's PCM is the first half of its 4608-byte place (include/pdmp3_hip.h); the other half is defined as
    // zero in the gathered output -- the shard's buffer is reused from call to call, so it is cleared when the shard has one
    bool mono = false;
    for (long long f = r.first; f < r.first + r.count && !mono; f++) mono = ((flags[(size_t)f] & PDMP3_FR_MODE_MASK) >> PDMP3_FR_MODE_SHIFT) == 3;
    if (mono && hipMemsetAsync(r.d_pcm, 0, (size_t)r.count * PDMP3_FRAME_PCM_BYTES, r.stream) != hipSuccess) {
      r.rc = PDMP3_HIP_EDEVICE; snprintf(r.err, sizeof r.err, "pdmp3_node: hipMemsetAsync on device %d failed", r.device); return;
    }
    if (hipMemcpyAsync(r.d_spectra, spectra + (size_t)r.first * 2304, (size_t)r.count * PDMP3_FRAME_SPECTRA_BYTES, hipMemcpyHostToDevice, r.stream) != hipSuccess ||
        hipMemcpyAsync(r.d_side, side + (size_t)r.first * 4, (size_t)r.count * PDMP3_FRAME_SIDE_BYTES, hipMemcpyHostToDevice, r.stream) != hipSuccess ||
        hipStreamSynchronize(r.stream) != hipSuccess) {
      r.rc = PDMP3_HIP_EDEVICE; snprintf(r.err, sizeof r.err, "pdmp3_node: upload to device %d failed", r.device);
    }
  });
  if (t) t->prepare_ms = now_ms() - t0;
  if (rc == PDMP3_HIP_OK) rc = decode_and_gather(node, d_pcm, t);
  return rc;
}

extern "C" int pdmp3_node_decode_generated(pdmp3_node* node, uint64_t seed, long long n_frames, int16_t* d_pcm, pdmp3_node_timing* t) {
  if (!node || n_frames < 0 || (n_frames && !d_pcm)) return node_fail(PDMP3_HIP_EINVAL, "pdmp3_node_decode_generated: bad argument");
  if (t) memset(t, 0, sizeof *t);
  if (!n_frames) return PDMP3_HIP_OK;
  DeviceGuard guard;
  if (n_frames / node->n + 8 > 0x7fffffffLL) return node_fail(PDMP3_HIP_EINVAL, "pdmp3_node_decode_generated: shards of more than 2^31 frames");
  for (int k = 0; k < node->n; k++) {
    Rank& r = node->rank[(size_t)k];
    pdmp3_node_shard(n_frames, k, node->n, nullptr, &r.first, &r.count, &r.discard);      // (the generated stream is all stereo: the fixed halo)
    r.lo = r.first + r.discard;
  }
  const double t0 = now_ms();
  int rc = on_all_ranks(node, [&](int, Rank& r) {
    if (r.count <= 0) return;
    if ((r.rc = rank_reserve(r, r.count)) != PDMP3_HIP_OK) return;
    r.rc = pdmp3_hip_generate_frames(r.ctx, seed, (int64_t)r.first, (int)r.count, r.d_spectra, r.d_side, r.stream);
    if (r.rc == PDMP3_HIP_OK && hipStreamSynchronize(r.stream) != hipSuccess) { r.rc = PDMP3_HIP_EDEVICE; snprintf(r.err, sizeof r.err, "pdmp3_node: generation on device %d failed", r.device); }
  });
  if (t) t->prepare_ms = now_ms() - t0;
  if (rc == PDMP3_HIP_OK) rc = decode_and_gather(node, d_pcm, t);
  return rc;
}
