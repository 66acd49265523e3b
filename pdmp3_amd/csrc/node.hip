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
  hipStream_t stream = nullptr;
  ncclComm_t comm = nullptr;
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
  for (Rank& r : node->rank) {
    (void)hipSetDevice(r.device);
    if (r.stream) (void)hipStreamSynchronize(r.stream);
    if (r.comm && node->rccl.CommDestroy) (void)node->rccl.CommDestroy(r.comm);
    (void)hipFree(r.d_spectra); (void)hipFree(r.d_side); (void)hipFree(r.d_pcm);
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
  pdmp3_node* node = new pdmp3_node;
  node->n = n; node->transport = transport;
  node->rank.resize((size_t)n);
  for (int i = 0; i < n; i++) {
    Rank& r = node->rank[(size_t)i];
    r.device = devices[i];
    if (pdmp3_hip_create(r.device, &r.ctx) != PDMP3_HIP_OK) { pdmp3_node_destroy(node); return PDMP3_HIP_EDEVICE; }   // (text: the engine's)
    if (hipSetDevice(r.device) != hipSuccess || hipStreamCreateWithFlags(&r.stream, hipStreamNonBlocking) != hipSuccess) {
      pdmp3_node_destroy(node);
      return node_fail(PDMP3_HIP_EDEVICE, "pdmp3_node_create: hipStreamCreate failed");
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

// the exchange: rank k's frames [lo, hi) -- its PCM behind the discarded halo -- into d_pcm + lo on rank 0's device
int gather(pdmp3_node* node, int16_t* d_pcm, pdmp3_node_timing* t) {
  const double t0 = now_ms();
  long long moved = 0;
  Rank& root = node->rank[0];
  auto src_of = [](Rank& r) { return (const char*)r.d_pcm + (size_t)r.discard * PDMP3_FRAME_PCM_BYTES; };
  auto bytes_of = [](Rank& r) { return (size_t)(r.count - r.discard) * PDMP3_FRAME_PCM_BYTES; };
  auto dst_of = [&](Rank& r) { return (char*)d_pcm + (size_t)r.lo * PDMP3_FRAME_PCM_BYTES; };
  // rank 0's own share never leaves its device
  if (hipSetDevice(root.device) != hipSuccess ||
      hipMemcpyAsync(dst_of(root), src_of(root), bytes_of(root), hipMemcpyDeviceToDevice, root.stream) != hipSuccess)
    return node_fail(PDMP3_HIP_EDEVICE, "pdmp3_node: copy of rank 0's PCM failed");
  if (node->transport == PDMP3_NODE_RCCL && node->n > 1) {
    // one group: every sender's ncclSend on its own communicator and stream, the root's ncclRecv for each of them --
    // point-to-point transfers that run side by side on the root's ingress links (RCCL has no int16: bytes)
    const Rccl& R = node->rccl;
    ncclResult_t rc = R.GroupStart();
    for (int k = 1; k < node->n && rc == ncclSuccess; k++) {
      Rank& r = node->rank[(size_t)k];
      if (!bytes_of(r)) continue;
      rc = R.Send(src_of(r), bytes_of(r), ncclChar, 0, r.comm, r.stream);
      if (rc == ncclSuccess) rc = R.Recv(dst_of(r), bytes_of(r), ncclChar, k, root.comm, root.stream);
      moved += (long long)bytes_of(r);
    }
    const ncclResult_t rc2 = R.GroupEnd();
    if (rc != ncclSuccess || rc2 != ncclSuccess)
      return node_fail(PDMP3_HIP_EDEVICE, "pdmp3_node: RCCL gather: %s", R.GetErrorString(rc != ncclSuccess ? rc : rc2));
  } else {
    for (int k = 1; k < node->n; k++) {
      Rank& r = node->rank[(size_t)k];
      if (!bytes_of(r)) continue;
      // (ordered behind the rank's decode: enqueued on ITS stream; the destination belongs to the root's device)
      if (hipSetDevice(r.device) != hipSuccess ||
          hipMemcpyPeerAsync(dst_of(r), root.device, src_of(r), r.device, bytes_of(r), r.stream) != hipSuccess)
        return node_fail(PDMP3_HIP_EDEVICE, "pdmp3_node: hipMemcpyPeerAsync failed");
      moved += (long long)bytes_of(r);
    }
  }
  for (Rank& r : node->rank) {
    if (hipSetDevice(r.device) != hipSuccess || hipStreamSynchronize(r.stream) != hipSuccess)
      return node_fail(PDMP3_HIP_EDEVICE, "pdmp3_node: synchronising the gather failed");
  }
  if (t) { t->gather_ms = now_ms() - t0; t->gather_bytes = moved; t->rccl_ranks = node->rccl_ranks; }
  return PDMP3_HIP_OK;
}

int decode_all(pdmp3_node* node, pdmp3_node_timing* t) {
  const double t0 = now_ms();
  const int rc = on_all_ranks(node, [](int, Rank& r) {
    if (r.count <= 0) return;
    // (one pdmp3_hip_decode_frames call takes an int: shards of more than 2^31 - 1 frames do not exist)
    r.rc = pdmp3_hip_decode_frames(r.ctx, r.d_spectra, r.d_side, (int)r.count, nullptr, r.d_pcm, 0, r.stream);
    if (r.rc == PDMP3_HIP_OK && hipStreamSynchronize(r.stream) != hipSuccess) { r.rc = PDMP3_HIP_EDEVICE; snprintf(r.err, sizeof r.err, "pdmp3_node: decode on device %d failed", r.device); }
  });
  if (t) t->decode_ms = now_ms() - t0;
  return rc;
}

}  // namespace

extern "C" int pdmp3_node_decode_records(pdmp3_node* node, const int16_t* spectra, const pdmp3_gc_side* side, long long n_frames,
                                         int16_t* d_pcm, pdmp3_node_timing* t) {
  if (!node || n_frames < 0 || (n_frames && (!spectra || !side || !d_pcm))) return node_fail(PDMP3_HIP_EINVAL, "pdmp3_node_decode_records: bad argument");
  if (t) memset(t, 0, sizeof *t);
  if (!n_frames) return PDMP3_HIP_OK;
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
  if (rc == PDMP3_HIP_OK) rc = decode_all(node, t);
  if (rc == PDMP3_HIP_OK) rc = gather(node, d_pcm, t);
  return rc;
}

extern "C" int pdmp3_node_decode_generated(pdmp3_node* node, uint64_t seed, long long n_frames, int16_t* d_pcm, pdmp3_node_timing* t) {
  if (!node || n_frames < 0 || (n_frames && !d_pcm)) return node_fail(PDMP3_HIP_EINVAL, "pdmp3_node_decode_generated: bad argument");
  if (t) memset(t, 0, sizeof *t);
  if (!n_frames) return PDMP3_HIP_OK;
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
  if (rc == PDMP3_HIP_OK) rc = decode_all(node, t);
  if (rc == PDMP3_HIP_OK) rc = gather(node, d_pcm, t);
  return rc;
}
