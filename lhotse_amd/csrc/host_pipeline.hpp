// The offline driver's extraction step as ONE asynchronous native call per batch (the caller is
// compute_and_store_features_batch's main thread, lhotse/cut/set.py:2365-2404): host waveforms of a batch -> a few host threads pack
// them into page-locked staging (a persistent pool: no per-batch thread start, no interpreter) -> H2D, [int16 -> float32,] the plan's
// feature launch, [float32 -> binary16,] D2H into a page-locked result, chunk by chunk on two streams, so that the upload of chunk
// n + 1, the launches of chunk n and the download of chunk n - 1 overlap -- and, because submit() only ENQUEUES, so do the tail of
// batch n and the packing of batch n + 1.  wait(ticket) blocks until the batch's features are on the host; release(ticket) hands the
// result buffer back.  Included at the end of hipfeat.hip (uses its fail / HIP_TRY / DeviceGuard and the extern "C" entry points).
#pragma once

#include <atomic>
#include <condition_variable>
#include <thread>

namespace {

struct CopyTask {
  char* dst;
  const char* src;
  size_t bytes;
};

// A few persistent host threads that run batches of memcpy tasks; the submitting thread takes its share.
class CopyPool {
 public:
  explicit CopyPool(int workers) {
    for (int i = 0; i < workers; ++i) th_.emplace_back([this] { loop(); });
  }
  ~CopyPool() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto& t : th_) t.join();
  }
  void run(const std::vector<CopyTask>& tasks) {
    if (tasks.empty()) return;
    if (tasks.size() == 1 || th_.empty()) {
      for (const auto& t : tasks) std::memcpy(t.dst, t.src, t.bytes);
      return;
    }
    {
      std::lock_guard<std::mutex> lk(mu_);
      tasks_ = &tasks;
      next_.store(0, std::memory_order_relaxed);
      active_ = (int)th_.size();
      ++gen_;
    }
    cv_.notify_all();
    drain(tasks);
    std::unique_lock<std::mutex> lk(mu_);
    done_.wait(lk, [this] { return active_ == 0; });
    tasks_ = nullptr;
  }

 private:
  void drain(const std::vector<CopyTask>& tasks) {
    for (;;) {
      const size_t i = next_.fetch_add(1, std::memory_order_relaxed);
      if (i >= tasks.size()) return;
      std::memcpy(tasks[i].dst, tasks[i].src, tasks[i].bytes);
    }
  }
  void loop() {
    uint64_t seen = 0;
    for (;;) {
      const std::vector<CopyTask>* t;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
        if (stop_) return;
        seen = gen_;
        t = tasks_;
      }
      drain(*t);
      {
        std::lock_guard<std::mutex> lk(mu_);
        if (--active_ == 0) done_.notify_one();
      }
    }
  }
  std::vector<std::thread> th_;
  std::mutex mu_;
  std::condition_variable cv_, done_;
  const std::vector<CopyTask>* tasks_ = nullptr;
  std::atomic<size_t> next_{0};
  int active_ = 0;
  uint64_t gen_ = 0;
  bool stop_ = false;
};

constexpr size_t kPipeCopyPiece = (size_t)1 << 20;        // bytes per memcpy task
constexpr int64_t kPipeMinChunkBytes = (int64_t)4 << 20;  // a chunk = consecutive cuts of at least this many input bytes ...
constexpr int kPipeTargetChunks = 4;                      // ... about this many per batch
constexpr int kPipeInSlots = 3;                           // input staging sets in rotation (a batch may be submitted while two are in flight)

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
  hipEvent_t uploaded = nullptr;    // the last H2D copy out of `h` has finished: `h` may be packed again
  hipEvent_t downloaded = nullptr;  // the last D2H copy out of d_feat / d_half has finished: they may be written again
  bool used = false;
};

struct PipeOut {  // one page-locked result buffer
  void* h = nullptr;
  size_t cap = 0;
  hipEvent_t done = nullptr;
  int64_t ticket = -1;  // -1 = free
};

template <typename T>
hipError_t grow_device(T** p, size_t* cap, size_t need_bytes) {
  if (*cap >= need_bytes) return hipSuccess;
  if (*p) (void)hipFree(*p);
  *p = nullptr;
  *cap = 0;
  const size_t want = need_bytes + need_bytes / 4;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(p), want);
  if (e == hipSuccess) *cap = want;
  return e;
}

}  // namespace

struct hipfeat_host_pipeline {
  const hipfeat_plan* plan = nullptr;
  int device = 0;
  hipStream_t s_in = nullptr, s_out = nullptr;
  CopyPool* pool = nullptr;
  PipeIn in[kPipeInSlots];
  std::vector<PipeOut> outs;
  int64_t next_ticket = 0;
  std::mutex mu;
};

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
  p->pool = new CopyPool(copy_threads - 1);  // the submitting thread copies too
  *out = p;
  return HIPFEAT_OK;
}

extern "C" HIPFEAT_API hipfeat_status hipfeat_host_pipeline_destroy(hipfeat_host_pipeline* p) {
  if (!p) return HIPFEAT_OK;
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
  // where every cut goes, its frame count, the chunks
  std::vector<int64_t> off((size_t)batch), frames((size_t)batch), row0((size_t)batch + 1, 0), padded;
  int64_t total = 0, max_len = 0;
  for (int64_t b = 0; b < batch; ++b) {
    const int64_t n = h_num_samples[b];
    if (n < 0 || n > INT32_MAX / 2 || (n > 0 && !h_items[b])) return fail(HIPFEAT_ERR_INVALID, "cut %lld: bad pointer / %lld samples", (long long)b, (long long)n);
    off[(size_t)b] = total;
    total += (n + align - 1) & ~(align - 1);
    max_len = std::max(max_len, n);
  }
  if (zero_pad_batch) padded.assign((size_t)batch, max_len);
  for (int64_t b = 0; b < batch; ++b) {
    int64_t T = hipfeat_num_frames(h_num_samples[b], c.frame_length, c.frame_shift, c.snip_edges);
    if (zero_pad_batch) {
      const int64_t hop = c.batch_hop > 0 ? c.batch_hop : c.frame_shift;
      T = std::min<int64_t>((h_num_samples[b] + hop / 2) / hop, hipfeat_num_frames(max_len, c.frame_length, c.frame_shift, c.snip_edges));
    }
    if (T <= 0) return fail(HIPFEAT_ERR_TOO_SHORT, "cut %lld: %lld samples give no frame", (long long)b, (long long)h_num_samples[b]);
    frames[(size_t)b] = T;
    row0[(size_t)b + 1] = row0[(size_t)b] + T;
    if (h_num_frames) h_num_frames[b] = T;
  }
  const int64_t rows = row0[(size_t)batch];
  const size_t out_item = half_out ? 2 : 4;
  const size_t out_bytes = (size_t)rows * F * out_item;
  std::vector<std::pair<int64_t, int64_t>> chunks;
  {
    const int64_t target = std::max<int64_t>(kPipeMinChunkBytes, (int64_t)(total * in_item) / kPipeTargetChunks);
    int64_t a = 0, acc = 0;
    for (int64_t b = 0; b < batch; ++b) {
      acc += ((h_num_samples[b] + align - 1) & ~(align - 1)) * (int64_t)in_item;
      if (acc >= target) {
        chunks.emplace_back(a, b + 1);
        a = b + 1;
        acc = 0;
      }
    }
    if (a < batch) chunks.emplace_back(a, batch);
  }

  std::lock_guard<std::mutex> lk(p->mu);
  DeviceGuard g(p->device);
  const int64_t tk = p->next_ticket;
  PipeIn& s = p->in[tk % kPipeInSlots];
  // the staging set's previous batch: its uploads must have left `h`; its downloads are ordered on the device (below)
  if (s.used) HIP_TRY(hipEventSynchronize(s.uploaded));
  const size_t in_bytes = (size_t)total * in_item;
  if (s.h_cap < in_bytes) {
    if (s.h) (void)hipHostFree(s.h);
    s.h = nullptr;
    s.h_cap = 0;
    const size_t want = in_bytes + in_bytes / 4;
    HIP_TRY(hipHostMalloc(&s.h, want, hipHostMallocDefault));
    s.h_cap = want;
  }
  HIP_TRY(grow_device(&s.d_raw, &s.raw_cap, in_bytes));
  if (pcm16) HIP_TRY(grow_device(&s.d_wave, &s.wave_cap, (size_t)total * 4));
  HIP_TRY(grow_device(&s.d_feat, &s.feat_cap, (size_t)rows * F * 4));
  if (half_out) HIP_TRY(grow_device(&s.d_half, &s.half_cap, (size_t)rows * F * 2));
  // a free result buffer (the smallest one that is large enough; else a new one)
  int oi = -1;
  for (size_t i = 0; i < p->outs.size(); ++i)
    if (p->outs[i].ticket < 0 && p->outs[i].cap >= out_bytes && (oi < 0 || p->outs[i].cap < p->outs[(size_t)oi].cap)) oi = (int)i;
  if (oi < 0) {
    for (size_t i = 0; i < p->outs.size() && oi < 0; ++i)
      if (p->outs[i].ticket < 0) {  // too small: replace it
        (void)hipHostFree(p->outs[i].h);
        p->outs[i].h = nullptr;
        p->outs[i].cap = 0;
        oi = (int)i;
      }
    if (oi < 0) {
      if (p->outs.size() >= 64) return fail(HIPFEAT_ERR_INVALID, "host pipeline: 64 results are outstanding: release finished batches");
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

  // the device buffers of this set are being read by the D2H copies of its previous batch until `downloaded`
  if (s.used) HIP_TRY(hipStreamWaitEvent(p->s_in, s.downloaded, 0));
  char* hin = static_cast<char*>(s.h);
  float* d_wave = pcm16 ? s.d_wave : static_cast<float*>(s.d_raw);
  std::vector<CopyTask> tasks;
  std::vector<int64_t> c_off, c_rows;
  hipfeat_status st = HIPFEAT_OK;
  while (s.chunk_done.size() < chunks.size()) {
    hipEvent_t e = nullptr;
    HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    s.chunk_done.push_back(e);
  }
  size_t ci = 0;
  for (const auto& ch : chunks) {
    const int64_t a = ch.first, b = ch.second;
    tasks.clear();
    for (int64_t i = a; i < b; ++i) {
      const char* src = static_cast<const char*>(h_items[i]);
      char* dst = hin + (size_t)off[(size_t)i] * in_item;
      const size_t nbytes = (size_t)h_num_samples[i] * in_item;
      for (size_t q = 0; q < nbytes; q += kPipeCopyPiece) tasks.push_back(CopyTask{dst + q, src + q, std::min(kPipeCopyPiece, nbytes - q)});
    }
    p->pool->run(tasks);
    const int64_t e0 = off[(size_t)a], e1 = (b < batch ? off[(size_t)b] : total);
    HIP_TRY(hipMemcpyAsync(static_cast<char*>(s.d_raw) + (size_t)e0 * in_item, hin + (size_t)e0 * in_item, (size_t)(e1 - e0) * in_item, hipMemcpyHostToDevice, p->s_in));
    if (pcm16) {
      st = hipfeat_pcm16_to_float(static_cast<const int16_t*>(s.d_raw) + e0, s.d_wave + e0, e1 - e0, p->s_in);
      if (st != HIPFEAT_OK) return st;
    }
    c_off.assign(off.begin() + a, off.begin() + b);
    c_rows.assign(row0.begin() + a, row0.begin() + b);
    st = hipfeat_extract(plan, d_wave, c_off.data(), h_num_samples + a, zero_pad_batch ? padded.data() + a : nullptr, b - a, s.d_feat, c_rows.data(), F, p->s_in);
    if (st != HIPFEAT_OK) return st;
    const int64_t r0 = row0[(size_t)a], r1 = row0[(size_t)b];
    const void* src_dev = s.d_feat + r0 * F;
    if (half_out) {
      st = hipfeat_float_to_half(s.d_feat + r0 * F, s.d_half + r0 * F, (r1 - r0) * F, p->s_in);
      if (st != HIPFEAT_OK) return st;
      src_dev = s.d_half + r0 * F;
    }
    // chunk finished on s_in -> its download on s_out
    hipEvent_t ev_chunk = s.chunk_done[ci++];
    HIP_TRY(hipEventRecord(ev_chunk, p->s_in));
    HIP_TRY(hipStreamWaitEvent(p->s_out, ev_chunk, 0));
    HIP_TRY(hipMemcpyAsync(static_cast<char*>(o.h) + (size_t)r0 * F * out_item, src_dev, (size_t)(r1 - r0) * F * out_item, hipMemcpyDeviceToHost, p->s_out));
  }
  HIP_TRY(hipEventRecord(s.uploaded, p->s_in));
  HIP_TRY(hipEventRecord(s.downloaded, p->s_out));
  HIP_TRY(hipEventRecord(o.done, p->s_out));
  s.used = true;
  o.ticket = tk;
  p->next_ticket = tk + 1;
  *h_out = o.h;
  if (h_out_rows) *h_out_rows = rows;
  *ticket = tk;
  return HIPFEAT_OK;
}

static PipeOut* pipe_find(hipfeat_host_pipeline* p, int64_t ticket) {
  for (auto& o : p->outs)
    if (o.ticket == ticket) return &o;
  return nullptr;
}

extern "C" HIPFEAT_API hipfeat_status hipfeat_host_pipeline_wait(hipfeat_host_pipeline* p, int64_t ticket) {
  if (!p) return fail(HIPFEAT_ERR_INVALID, "pipeline is NULL");
  hipEvent_t ev;
  {
    std::lock_guard<std::mutex> lk(p->mu);
    PipeOut* o = pipe_find(p, ticket);
    if (!o) return fail(HIPFEAT_ERR_INVALID, "host pipeline: ticket %lld is not outstanding", (long long)ticket);
    ev = o->done;
  }
  DeviceGuard g(p->device);
  HIP_TRY(hipEventSynchronize(ev));  // (outside the lock: other threads keep submitting)
  return HIPFEAT_OK;
}

extern "C" HIPFEAT_API hipfeat_status hipfeat_host_pipeline_release(hipfeat_host_pipeline* p, int64_t ticket) {
  if (!p) return fail(HIPFEAT_ERR_INVALID, "pipeline is NULL");
  std::lock_guard<std::mutex> lk(p->mu);
  PipeOut* o = pipe_find(p, ticket);
  if (!o) return fail(HIPFEAT_ERR_INVALID, "host pipeline: ticket %lld is not outstanding", (long long)ticket);
  DeviceGuard g(p->device);
  HIP_TRY(hipEventSynchronize(o->done));  // a buffer is never handed out again while its download is in flight
  o->ticket = -1;
  return HIPFEAT_OK;
}
