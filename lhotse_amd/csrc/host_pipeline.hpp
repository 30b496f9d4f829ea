// The offline driver's extraction step as ONE asynchronous native call per batch (the caller is
// compute_and_store_features_batch's main thread, lhotse/cut/set.py:2365-2404).
//
//   submit()   caller's thread: where every cut goes, frame counts, a page-locked result buffer, a ticket -- and the batch is QUEUED.
//   worker     one pipeline thread per pipeline takes the batches in order: a persistent pool of host threads packs the cuts into
//              page-locked staging, and the upload, [int16 -> float32,] the plan's feature launch, [float32 -> binary16,] and the
//              download into the result buffer are enqueued chunk by chunk on two streams (upload of chunk n + 1, launches of
//              chunk n, download of chunk n - 1 overlap; so do the tail of batch n and the packing of batch n + 1).
//   wait()     blocks until the batch has been enqueued and its download has finished;  release() hands the result buffer back.
//
// The calling (interpreter) thread thus spends microseconds per batch; packing, PCIe and the save threads run beside it.
// Included at the end of hipfeat.hip (uses its fail / HIP_TRY / DeviceGuard and the extern "C" entry points).
#pragma once

#include <chrono>
#include <deque>
#include <map>
#include <memory>

namespace {

constexpr size_t kPipeCopyPiece = (size_t)1 << 20;        // bytes per memcpy task
constexpr int64_t kPipeMinChunkBytes = (int64_t)4 << 20;  // a chunk = consecutive cuts of at least this many input bytes ...
// ... about this many per batch (HIPFEAT_PIPE_CHUNKS: routing switch, 1 ... 16).  ONE: with submit() asynchronous the overlap that matters is
// between batches (batch n + 1 is packed and uploaded while batch n downloads), and every chunk costs the pipeline thread ~0.1 ms of
// enqueue calls: same-call A/B on the offline path, 1 vs 4 chunks: + 3 ... 30 % in 8 of 8 comparisons (profiles/r05_pipeline_chunks_ab.txt)
constexpr int kPipeTargetChunks = 1;
constexpr int kPipeInSlots = 3;                           // input staging sets in rotation
constexpr size_t kPipeMaxOutstanding = 64;

struct PipeIn {  // one input staging set: pinned host buffer + the device buffers of a batch in flight
  void* h = nullptr;
  size_t h_cap = 0;
  void* d_raw = nullptr;  // what was uploaded (float32 samples, or int16 PCM)
  size_t raw_cap = 0;
  float* d_wave = nullptr;  // float32 samples when the upload was PCM
  size_t wave_cap = 0;
  float* d_feat = nullptr;
  size_t feat_cap = 0;
  uint16_t* d_half = nullptr;
  size_t half_cap = 0;
  std::vector<hipEvent_t> chunk_done;  // chunk c's launches have finished on s_in (one event per chunk of the batch in flight)
  hipEvent_t uploaded = nullptr;       // the last H2D copy out of `h` has finished: `h` may be packed again
  hipEvent_t downloaded = nullptr;     // the last D2H copy out of d_feat / d_half has finished: they may be written again
  bool used = false;
};

struct PipeOut {  // one page-locked result buffer
  void* h = nullptr;
  size_t cap = 0;
  hipEvent_t done = nullptr;
  int64_t ticket = -1;  // -1 = free
};

// memory of the caller that is page-locked for DMA (hipfeat_host_register): base -> (bytes, device)
struct PinnedRange {
  size_t bytes;
  int device;
};
std::mutex g_pinned_mu;
std::map<const char*, PinnedRange> g_pinned;

// the registered range [p, p + bytes) lies in, for `device`: its base, or nullptr
const char* pinned_base_of(const void* p, size_t bytes, int device) {
  std::lock_guard<std::mutex> lk(g_pinned_mu);
  const char* c = static_cast<const char*>(p);
  auto it = g_pinned.upper_bound(c);
  if (it == g_pinned.begin()) return nullptr;
  --it;
  if (it->second.device != device || c + bytes > it->first + it->second.bytes) return nullptr;
  return it->first;
}

struct PipeJob {
  int64_t ticket = 0;
  bool direct = false;  // the cuts lie back to back in page-locked memory of the caller: uploaded from there, no staging
  std::vector<const void*> items;
  std::vector<int64_t> lens, off, row0, padded;
  std::vector<std::pair<int64_t, int64_t>> chunks;
  int64_t total = 0, rows = 0;
  bool pcm16 = false, zero_pad = false, half = false;
  int out = -1;  // index into hipfeat_host_pipeline::outs
  // set by the worker
  bool enqueued = false;
  hipfeat_status status = HIPFEAT_OK;
  std::string error;
};

hipError_t grow_bytes(void** p, size_t* cap, size_t need_bytes) {
  if (*cap >= need_bytes) return hipSuccess;
  if (*p) (void)hipFree(*p);  // (synchronises with the device: nothing in flight uses it afterwards)
  *p = nullptr;
  *cap = 0;
  const size_t want = need_bytes + need_bytes / 4;
  hipError_t e = hipMalloc(p, want);
  if (e == hipSuccess) *cap = want;
  return e;
}

}  // namespace

struct hipfeat_host_pipeline {
  const hipfeat_plan* plan = nullptr;
  int device = 0;
  hipStream_t s_in = nullptr, s_out = nullptr;
  hipfeat::WorkPool* pool = nullptr;
  PipeIn in[kPipeInSlots];
  std::vector<PipeOut> outs;
  int64_t next_ticket = 0;
  std::mutex mu;  // outs, jobs, queue, next_ticket
  std::condition_variable cv_work, cv_done;
  std::deque<std::shared_ptr<PipeJob>> queue;
  std::map<int64_t, std::shared_ptr<PipeJob>> jobs;
  std::thread worker;
  bool stop = false;
  int target_chunks = kPipeTargetChunks;
  // the pipeline thread's own clock (hipfeat_host_pipeline_stats): nanoseconds busy with batches / of those: packing / of those: waiting
  // for a staging set's previous uploads and downloads (= back-pressure from PCIe and the device); batches processed
  std::atomic<int64_t> ns_busy{0}, ns_pack{0}, ns_slot_wait{0}, n_batches{0}, n_direct{0};
};

static inline int64_t pipe_now_ns() {
  return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// The worker's side of one batch: everything that touches the staging sets and the streams.  Returns a status; on failure the
// message is in this thread's g_err.
static hipfeat_status pipe_process_batch(hipfeat_host_pipeline* p, PipeJob& j, PipeOut o);

// A batch that fails half-way (a HIP error, a refused launch) has left part of its uploads / launches / downloads on the two streams and
// its staging set's events un-recorded: the streams are drained HERE, on the worker, and the set is marked idle, so that the batches
// queued behind it find a consistent slot (three tickets later the slot used to be reused after "synchronising" stale events).
static hipfeat_status pipe_process(hipfeat_host_pipeline* p, PipeJob& j, PipeOut o) {
  const hipfeat_status st = pipe_process_batch(p, j, o);
  if (st != HIPFEAT_OK) {
    const std::string why = g_err;  // (the synchronisation below must not replace the batch's own message)
    DeviceGuard g(p->device);
    (void)hipStreamSynchronize(p->s_in);
    (void)hipStreamSynchronize(p->s_out);
    p->in[j.ticket % kPipeInSlots].used = false;
    snprintf(g_err, sizeof(g_err), "%s", why.c_str());
  }
  return st;
}

static hipfeat_status pipe_process_batch(hipfeat_host_pipeline* p, PipeJob& j, PipeOut o) {
  const hipfeat_plan* plan = p->plan;
  const int F = plan->feature_dim;
  const int64_t batch = (int64_t)j.items.size();
  const size_t in_item = j.pcm16 ? 2 : 4, out_item = j.half ? 2 : 4;
  DeviceGuard g(p->device);
  PipeIn& s = p->in[j.ticket % kPipeInSlots];
  // the staging set's previous batch (three batches ago): its uploads must have left `h`, and its downloads must be over before the
  // chunk events they wait on are recorded again and before its device buffers are written
  if (s.used) {
    const int64_t t0 = pipe_now_ns();
    HIP_TRY(hipEventSynchronize(s.uploaded));
    HIP_TRY(hipEventSynchronize(s.downloaded));
    p->ns_slot_wait.fetch_add(pipe_now_ns() - t0, std::memory_order_relaxed);
  }
  const size_t in_bytes = (size_t)j.total * in_item;
  if (!j.direct && s.h_cap < in_bytes) {
    if (s.h) (void)hipHostFree(s.h);
    s.h = nullptr;
    s.h_cap = 0;
    const size_t want = in_bytes + in_bytes / 4;
    HIP_TRY(hipHostMalloc(&s.h, want, hipHostMallocDefault));
    s.h_cap = want;
  }
  HIP_TRY(grow_bytes(&s.d_raw, &s.raw_cap, in_bytes));
  if (j.pcm16) HIP_TRY(grow_bytes(reinterpret_cast<void**>(&s.d_wave), &s.wave_cap, (size_t)j.total * 4));
  HIP_TRY(grow_bytes(reinterpret_cast<void**>(&s.d_feat), &s.feat_cap, (size_t)j.rows * F * 4));
  if (j.half) HIP_TRY(grow_bytes(reinterpret_cast<void**>(&s.d_half), &s.half_cap, (size_t)j.rows * F * 2));
  while (s.chunk_done.size() < j.chunks.size()) {
    hipEvent_t e = nullptr;
    HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    s.chunk_done.push_back(e);
  }
  // direct: "staging" is the caller's own page-locked memory (element 0 = the first cut; j.off are the cuts' offsets in it)
  char* hin = j.direct ? const_cast<char*>(static_cast<const char*>(j.items[0])) : static_cast<char*>(s.h);
  float* d_wave = j.pcm16 ? s.d_wave : static_cast<float*>(s.d_raw);
  struct Piece {
    char* dst;
    const char* src;
    size_t bytes;
  };
  std::vector<Piece> pieces;
  std::vector<int64_t> c_off, c_rows;
  size_t ci = 0;
  for (const auto& ch : j.chunks) {
    const int64_t a = ch.first, b = ch.second;
    pieces.clear();
    for (int64_t i = a; i < b && !j.direct; ++i) {
      const char* src = static_cast<const char*>(j.items[(size_t)i]);
      char* dst = hin + (size_t)j.off[(size_t)i] * in_item;
      const size_t nbytes = (size_t)j.lens[(size_t)i] * in_item;
      for (size_t q = 0; q < nbytes; q += kPipeCopyPiece) pieces.push_back(Piece{dst + q, src + q, std::min(kPipeCopyPiece, nbytes - q)});
    }
    if (!j.direct) {
      const int64_t t0 = pipe_now_ns();
      p->pool->run(pieces.size(), [&](size_t i) { std::memcpy(pieces[i].dst, pieces[i].src, pieces[i].bytes); });
      p->ns_pack.fetch_add(pipe_now_ns() - t0, std::memory_order_relaxed);
    }
    const int64_t e0 = j.off[(size_t)a], e1 = (b < batch ? j.off[(size_t)b] : j.total);
    HIP_TRY(hipMemcpyAsync(static_cast<char*>(s.d_raw) + (size_t)e0 * in_item, hin + (size_t)e0 * in_item, (size_t)(e1 - e0) * in_item, hipMemcpyHostToDevice, p->s_in));
    hipfeat_status st;
    if (j.pcm16) {
      st = hipfeat_pcm16_to_float(static_cast<const int16_t*>(s.d_raw) + e0, s.d_wave + e0, e1 - e0, p->s_in);
      if (st != HIPFEAT_OK) return st;
    }
    c_off.assign(j.off.begin() + a, j.off.begin() + b);
    c_rows.assign(j.row0.begin() + a, j.row0.begin() + b);
    st = hipfeat_extract(plan, d_wave, c_off.data(), j.lens.data() + a, j.zero_pad ? j.padded.data() + a : nullptr, b - a, s.d_feat, c_rows.data(), F, p->s_in);
    if (st != HIPFEAT_OK) return st;
    const int64_t r0 = j.row0[(size_t)a], r1 = j.row0[(size_t)b];
    const void* src_dev = s.d_feat + r0 * F;
    if (j.half) {
      st = hipfeat_float_to_half(s.d_feat + r0 * F, s.d_half + r0 * F, (r1 - r0) * F, p->s_in);
      if (st != HIPFEAT_OK) return st;
      src_dev = s.d_half + r0 * F;
    }
    hipEvent_t ev_chunk = s.chunk_done[ci++];  // chunk finished on s_in -> its download on s_out
    HIP_TRY(hipEventRecord(ev_chunk, p->s_in));
    HIP_TRY(hipStreamWaitEvent(p->s_out, ev_chunk, 0));
    HIP_TRY(hipMemcpyAsync(static_cast<char*>(o.h) + (size_t)r0 * F * out_item, src_dev, (size_t)(r1 - r0) * F * out_item, hipMemcpyDeviceToHost, p->s_out));
  }
  HIP_TRY(hipEventRecord(s.uploaded, p->s_in));
  HIP_TRY(hipEventRecord(s.downloaded, p->s_out));
  HIP_TRY(hipEventRecord(o.done, p->s_out));
  s.used = true;
  if (j.direct) p->n_direct.fetch_add(1, std::memory_order_relaxed);
  return HIPFEAT_OK;
}

static void pipe_worker(hipfeat_host_pipeline* p) {
  for (;;) {
    std::shared_ptr<PipeJob> j;
    PipeOut o;
    {
      std::unique_lock<std::mutex> lk(p->mu);
      p->cv_work.wait(lk, [p] { return p->stop || !p->queue.empty(); });
      if (p->queue.empty()) return;  // (stop, and nothing left to do)
      j = p->queue.front();
      p->queue.pop_front();
      o = p->outs[(size_t)j->out];  // a copy: `outs` may grow while this batch is processed; its buffer and event do not move
    }
    const int64_t t0 = pipe_now_ns();
    const hipfeat_status st = pipe_process(p, *j, o);
    p->ns_busy.fetch_add(pipe_now_ns() - t0, std::memory_order_relaxed);
    p->n_batches.fetch_add(1, std::memory_order_relaxed);
    {
      std::lock_guard<std::mutex> lk(p->mu);
      j->status = st;
      if (st != HIPFEAT_OK) j->error = g_err;
      j->items.clear();  // the caller's waveforms are not touched after this point
      j->enqueued = true;
    }
    p->cv_done.notify_all();
  }
}

extern "C" HIPFEAT_API hipfeat_status hipfeat_host_pipeline_create(const hipfeat_plan* plan, int32_t copy_threads, hipfeat_host_pipeline** out) {
  if (!out) return fail(HIPFEAT_ERR_INVALID, "pipeline pointer is NULL");
  *out = nullptr;
  if (!plan) return fail(HIPFEAT_ERR_INVALID, "plan is NULL");
  if (copy_threads < 1 || copy_threads > 64) return fail(HIPFEAT_ERR_INVALID, "copy_threads must be 1 ... 64");
  DeviceGuard g(plan->device);
  hipfeat_host_pipeline* p = new (std::nothrow) hipfeat_host_pipeline();
  if (!p) return fail(HIPFEAT_ERR_INVALID, "out of host memory");
  p->plan = plan;
  p->device = plan->device;
  hipError_t e = hipStreamCreateWithFlags(&p->s_in, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&p->s_out, hipStreamNonBlocking);
  for (int i = 0; i < kPipeInSlots && e == hipSuccess; ++i) {
    e = hipEventCreateWithFlags(&p->in[i].uploaded, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&p->in[i].downloaded, hipEventDisableTiming);
  }
  if (e != hipSuccess) {
    for (auto& s : p->in) {
      if (s.uploaded) (void)hipEventDestroy(s.uploaded);
      if (s.downloaded) (void)hipEventDestroy(s.downloaded);
    }
    if (p->s_in) (void)hipStreamDestroy(p->s_in);
    if (p->s_out) (void)hipStreamDestroy(p->s_out);
    delete p;
    return fail(HIPFEAT_ERR_HIP, "host pipeline: stream / event creation failed: %s", hipGetErrorName(e));
  }
  if (const char* tc = route_env("HIPFEAT_PIPE_CHUNKS")) p->target_chunks = std::min(16, std::max(1, atoi(tc)));
  p->pool = new hipfeat::WorkPool(copy_threads - 1, "hipfeat-pack");  // the pipeline thread copies too
  p->worker = std::thread(pipe_worker, p);
  (void)pthread_setname_np(p->worker.native_handle(), "hipfeat-pipe");
  *out = p;
  return HIPFEAT_OK;
}

extern "C" HIPFEAT_API hipfeat_status hipfeat_host_pipeline_destroy(hipfeat_host_pipeline* p) {
  if (!p) return HIPFEAT_OK;
  {
    std::lock_guard<std::mutex> lk(p->mu);
    p->stop = true;  // (the worker still drains what is queued: the callers' buffers of those batches stay valid until wait / release)
  }
  p->cv_work.notify_all();
  if (p->worker.joinable()) p->worker.join();
  DeviceGuard g(p->device);
  (void)hipStreamSynchronize(p->s_in);
  (void)hipStreamSynchronize(p->s_out);
  delete p->pool;
  for (auto& s : p->in) {
    if (s.h) (void)hipHostFree(s.h);
    if (s.d_raw) (void)hipFree(s.d_raw);
    if (s.d_wave) (void)hipFree(s.d_wave);
    if (s.d_feat) (void)hipFree(s.d_feat);
    if (s.d_half) (void)hipFree(s.d_half);
    if (s.uploaded) (void)hipEventDestroy(s.uploaded);
    if (s.downloaded) (void)hipEventDestroy(s.downloaded);
    for (hipEvent_t e : s.chunk_done) (void)hipEventDestroy(e);
  }
  for (auto& o : p->outs) {
    if (o.h) (void)hipHostFree(o.h);
    if (o.done) (void)hipEventDestroy(o.done);
  }
  (void)hipStreamDestroy(p->s_in);
  (void)hipStreamDestroy(p->s_out);
  delete p;
  return HIPFEAT_OK;
}

extern "C" HIPFEAT_API hipfeat_status hipfeat_host_pipeline_submit(hipfeat_host_pipeline* p, const void* const* h_items, const int64_t* h_num_samples,
                                                                  int64_t batch, int32_t pcm16, int32_t zero_pad_batch, int32_t half_out,
                                                                  int64_t* h_num_frames, void** h_out, int64_t* h_out_rows, int64_t* ticket) {
  if (!p || !h_items || !h_num_samples || !h_out || !ticket || batch <= 0 || batch > 65535) return fail(HIPFEAT_ERR_INVALID, "host pipeline: bad arguments (1 ... 65535 cuts)");
  const hipfeat_plan* plan = p->plan;
  const hipfeat_config& c = plan->cfg;
  const int F = plan->feature_dim;
  const size_t in_item = pcm16 ? 2 : 4;
  const int64_t align = pcm16 ? 8 : 4;  // every cut starts on a 16-byte boundary of the staging buffer (and so of the device buffer)
  auto j = std::make_shared<PipeJob>();
  j->pcm16 = pcm16 != 0;
  j->zero_pad = zero_pad_batch != 0;
  j->half = half_out != 0;
  j->items.assign(h_items, h_items + batch);
  j->lens.assign(h_num_samples, h_num_samples + batch);
  j->off.resize((size_t)batch);
  j->row0.assign((size_t)batch + 1, 0);
  int64_t total = 0, max_len = 0;
  for (int64_t b = 0; b < batch; ++b) {
    const int64_t n = h_num_samples[b];
    if (n < 0 || n > INT32_MAX / 2 || (n > 0 && !h_items[b])) return fail(HIPFEAT_ERR_INVALID, "cut %lld: bad pointer / %lld samples", (long long)b, (long long)n);
    j->off[(size_t)b] = total;
    total += (n + align - 1) & ~(align - 1);
    max_len = std::max(max_len, n);
  }
  // The cuts as they lie in the caller's memory: ascending, each on a 16-byte boundary, with less slack between them than packing would
  // save, and all inside ONE range registered for this device -> the span is uploaded as it is.
  {
    const char* first = static_cast<const char*>(h_items[0]);
    bool ok = first && (reinterpret_cast<uintptr_t>(first) & 15) == 0;
    int64_t span = 0;  // elements from the first cut to the (aligned) end of the last
    for (int64_t b = 0; b < batch && ok; ++b) {
      const char* q = static_cast<const char*>(h_items[b]);
      const int64_t n = h_num_samples[b];
      if (n == 0) {
        ok = false;
        break;
      }
      const int64_t at = (q - first) / (int64_t)in_item;
      ok = q >= first && (reinterpret_cast<uintptr_t>(q) & 15) == 0 && at >= span;
      span = at + ((n + align - 1) & ~(align - 1));
    }
    if (ok) ok = span <= total + total / 16 + 64;
    // (the last cut's padding up to its 16-byte boundary is part of the upload: it must lie inside the registered range too)
    if (ok && pinned_base_of(first, (size_t)span * in_item, plan->device)) {
      j->direct = true;
      for (int64_t b = 0; b < batch; ++b) j->off[(size_t)b] = (static_cast<const char*>(h_items[b]) - first) / (int64_t)in_item;
      total = span;
    }
  }
  j->total = total;
  if (zero_pad_batch) j->padded.assign((size_t)batch, max_len);
  for (int64_t b = 0; b < batch; ++b) {
    int64_t T = hipfeat_num_frames(h_num_samples[b], c.frame_length, c.frame_shift, c.snip_edges);
    if (zero_pad_batch) {
      const int64_t hop = c.batch_hop > 0 ? c.batch_hop : c.frame_shift;
      T = std::min<int64_t>((h_num_samples[b] + hop / 2) / hop, hipfeat_num_frames(max_len, c.frame_length, c.frame_shift, c.snip_edges));
    }
    // what build_descs would refuse is refused here, at submit, not later on the pipeline thread
    if (!c.snip_edges && T > 0 && c.kind != HIPFEAT_WHISPER && c.kind != HIPFEAT_LIBROSA_FBANK) {
      const hipfeat_status cl = hipfeat_check_length(zero_pad_batch ? max_len : h_num_samples[b], c.frame_length, c.frame_shift, 0);
      if (cl != HIPFEAT_OK) return cl;
    }
    // (snip_edges: a cut shorter than one frame has ZERO rows, as in the reference (layers.py:745-746) and in build_descs -- not an error)
    if (T < 0 || (T == 0 && !c.snip_edges)) return fail(HIPFEAT_ERR_TOO_SHORT, "cut %lld: %lld samples yield no frames", (long long)b, (long long)h_num_samples[b]);
    j->row0[(size_t)b + 1] = j->row0[(size_t)b] + T;
    if (h_num_frames) h_num_frames[b] = T;
  }
  j->rows = j->row0[(size_t)batch];
  const size_t out_bytes = (size_t)j->rows * F * (half_out ? 2 : 4);
  {
    const int64_t target = std::max<int64_t>(kPipeMinChunkBytes, (int64_t)(total * in_item) / p->target_chunks);
    int64_t a = 0, acc = 0;
    for (int64_t b = 0; b < batch; ++b) {
      acc += ((h_num_samples[b] + align - 1) & ~(align - 1)) * (int64_t)in_item;
      if (acc >= target) {
        j->chunks.emplace_back(a, b + 1);
        a = b + 1;
        acc = 0;
      }
    }
    if (a < batch) j->chunks.emplace_back(a, batch);
  }

  std::lock_guard<std::mutex> lk(p->mu);
  if (p->stop) return fail(HIPFEAT_ERR_INVALID, "host pipeline: being destroyed");
  // a free result buffer: the smallest one that is large enough; else a too-small free one is replaced; else a new one
  int oi = -1;
  for (size_t i = 0; i < p->outs.size(); ++i)
    if (p->outs[i].ticket < 0 && p->outs[i].cap >= out_bytes && (oi < 0 || p->outs[i].cap < p->outs[(size_t)oi].cap)) oi = (int)i;
  if (oi < 0) {
    DeviceGuard g(p->device);
    for (size_t i = 0; i < p->outs.size() && oi < 0; ++i)
      if (p->outs[i].ticket < 0) {
        (void)hipHostFree(p->outs[i].h);
        p->outs[i].h = nullptr;
        p->outs[i].cap = 0;
        oi = (int)i;
      }
    if (oi < 0) {
      if (p->outs.size() >= kPipeMaxOutstanding) return fail(HIPFEAT_ERR_INVALID, "host pipeline: %d results are outstanding: release finished batches", (int)kPipeMaxOutstanding);
      p->outs.emplace_back();
      oi = (int)p->outs.size() - 1;
    }
    PipeOut& o = p->outs[(size_t)oi];
    if (!o.done) HIP_TRY(hipEventCreateWithFlags(&o.done, hipEventDisableTiming));
    const size_t want = std::max<size_t>(out_bytes + out_bytes / 8, 1 << 16);
    HIP_TRY(hipHostMalloc(&o.h, want, hipHostMallocDefault));
    o.cap = want;
  }
  PipeOut& o = p->outs[(size_t)oi];
  j->ticket = p->next_ticket++;
  j->out = oi;
  o.ticket = j->ticket;
  p->jobs[j->ticket] = j;
  p->queue.push_back(j);
  *h_out = o.h;
  if (h_out_rows) *h_out_rows = j->rows;
  *ticket = j->ticket;
  p->cv_work.notify_one();
  return HIPFEAT_OK;
}

// waits until the batch has been enqueued by the worker; -> its status (message copied into this thread's error slot)
static hipfeat_status pipe_await_enqueued(hipfeat_host_pipeline* p, int64_t ticket, hipEvent_t* done, bool* unknown = nullptr) {
  std::unique_lock<std::mutex> lk(p->mu);
  auto it = p->jobs.find(ticket);
  if (unknown) *unknown = it == p->jobs.end();
  if (it == p->jobs.end()) return fail(HIPFEAT_ERR_INVALID, "host pipeline: ticket %lld is not outstanding", (long long)ticket);
  std::shared_ptr<PipeJob> j = it->second;
  p->cv_done.wait(lk, [&] { return j->enqueued; });
  if (j->status != HIPFEAT_OK) return fail(j->status, "%s", j->error.c_str());
  *done = p->outs[(size_t)j->out].done;
  return HIPFEAT_OK;
}

extern "C" HIPFEAT_API hipfeat_status hipfeat_host_pipeline_wait(hipfeat_host_pipeline* p, int64_t ticket) {
  if (!p) return fail(HIPFEAT_ERR_INVALID, "pipeline is NULL");
  hipEvent_t ev = nullptr;
  hipfeat_status st = pipe_await_enqueued(p, ticket, &ev);
  if (st != HIPFEAT_OK) return st;
  DeviceGuard g(p->device);
  HIP_TRY(hipEventSynchronize(ev));  // (outside the lock: other threads keep submitting)
  return HIPFEAT_OK;
}

extern "C" HIPFEAT_API hipfeat_status hipfeat_host_pipeline_release(hipfeat_host_pipeline* p, int64_t ticket) {
  if (!p) return fail(HIPFEAT_ERR_INVALID, "pipeline is NULL");
  hipEvent_t ev = nullptr;
  bool unknown = false;
  hipfeat_status st = pipe_await_enqueued(p, ticket, &ev, &unknown);  // never before the worker is done with the caller's waveforms
  if (unknown) return st;  // (its own answer: nothing to give back)
  if (st == HIPFEAT_OK) {
    DeviceGuard g(p->device);
    (void)hipEventSynchronize(ev);  // a buffer is never handed out again while its download is in flight
  }  // (a batch that failed half-way was drained by the worker itself: pipe_process)
  std::lock_guard<std::mutex> lk(p->mu);
  auto it = p->jobs.find(ticket);
  if (it == p->jobs.end()) return fail(HIPFEAT_ERR_INVALID, "host pipeline: ticket %lld is not outstanding", (long long)ticket);
  p->outs[(size_t)it->second->out].ticket = -1;
  p->jobs.erase(it);
  return HIPFEAT_OK;
}

extern "C" HIPFEAT_API hipfeat_status hipfeat_host_register(int32_t device, void* ptr, int64_t bytes) {
  if (!ptr || bytes <= 0) return fail(HIPFEAT_ERR_INVALID, "host register: NULL pointer / %lld bytes", (long long)bytes);
  DeviceGuard g(device);
  {
    std::lock_guard<std::mutex> lk(g_pinned_mu);
    if (g_pinned.count(static_cast<const char*>(ptr))) return fail(HIPFEAT_ERR_INVALID, "host register: %p is registered already", ptr);
  }
  const hipError_t e = hipHostRegister(ptr, (size_t)bytes, hipHostRegisterDefault);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail(HIPFEAT_ERR_HIP, "hipHostRegister(%p, %lld bytes) failed: %s", ptr, (long long)bytes, hipGetErrorName(e));
  }
  std::lock_guard<std::mutex> lk(g_pinned_mu);
  g_pinned[static_cast<const char*>(ptr)] = PinnedRange{(size_t)bytes, (int)device};
  return HIPFEAT_OK;
}

extern "C" HIPFEAT_API hipfeat_status hipfeat_host_unregister(void* ptr) {
  int device = -1;
  {
    std::lock_guard<std::mutex> lk(g_pinned_mu);
    auto it = g_pinned.find(static_cast<const char*>(ptr));
    if (it == g_pinned.end()) return fail(HIPFEAT_ERR_INVALID, "host unregister: %p is not registered", ptr);
    device = it->second.device;
    g_pinned.erase(it);  // (first: no later submit takes the direct route out of it)
  }
  DeviceGuard g(device);
  (void)hipDeviceSynchronize();  // (teardown path: whatever the device still reads out of this memory finishes first)
  const hipError_t e = hipHostUnregister(ptr);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail(HIPFEAT_ERR_HIP, "hipHostUnregister(%p) failed: %s", ptr, hipGetErrorName(e));
  }
  return HIPFEAT_OK;
}

extern "C" HIPFEAT_API int64_t hipfeat_host_pipeline_direct_batches(const hipfeat_host_pipeline* p) {
  return p ? p->n_direct.load(std::memory_order_relaxed) : -1;
}

extern "C" HIPFEAT_API hipfeat_status hipfeat_host_pipeline_stats(const hipfeat_host_pipeline* p, int64_t* h_stats) {
  if (!p || !h_stats) return fail(HIPFEAT_ERR_INVALID, "NULL argument");
  h_stats[0] = p->ns_busy.load(std::memory_order_relaxed);
  h_stats[1] = p->ns_pack.load(std::memory_order_relaxed);
  h_stats[2] = p->ns_slot_wait.load(std::memory_order_relaxed);
  h_stats[3] = p->n_batches.load(std::memory_order_relaxed);
  return HIPFEAT_OK;
}
