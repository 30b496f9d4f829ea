// Host-side pieces of the bulk save path (lhotse/cut/set.py:2307-2363, lhotse/features/io.py:499-525): what the
// reference's _save_worker does per CUT in the interpreter -- store one matrix, build a Features object, validate its frame count,
// fastcopy the cut, to_dict, json.dumps -- restated per BATCH in plain C++, callable with the GIL released:
//   * manifest_lines: one JSONL line per cut = the two halves of the cut's own serialisation (made where the cut was loaded: the
//     loader's worker processes) spliced around the ONE field only the save path knows, the storage key, after the frame-count
//     contract of validate_features (lhotse/qa.py:286-301) has been checked for the whole batch;
//   * archive_append: the packed (sum T_b, F) matrix of a batch appended to the flat archive file(s) by a few writer threads.
// No HIP in this file.
#pragma once
#include <pthread.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <unistd.h>

namespace hipfeat {

// A few persistent host threads that run `n` indexed jobs per call; the calling thread takes its share.  (Starting a std::thread per
// batch costs 50-100 us each -- as much as the work itself for a 600 s batch cut eight ways.)
class WorkPool {
 public:
  // `name` (<= 15 characters) shows in /proc/<pid>/task/*/comm: per-thread CPU accounting of a run (tools/plumbing.py)
  explicit WorkPool(int workers, const char* name = "hipfeat-pool") {
    for (int i = 0; i < workers; ++i) {
      th_.emplace_back([this] { loop(); });
      (void)pthread_setname_np(th_.back().native_handle(), name);
    }
  }
  ~WorkPool() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto& t : th_) t.join();
  }
  WorkPool(const WorkPool&) = delete;
  WorkPool& operator=(const WorkPool&) = delete;
  int workers() const { return (int)th_.size(); }
  // fn(i) for every i in [0, n), spread over the workers and the caller; returns when all have finished.  One run at a time.
  void run(size_t n, const std::function<void(size_t)>& fn) {
    if (n == 0) return;
    if (n == 1 || th_.empty()) {
      for (size_t i = 0; i < n; ++i) fn(i);
      return;
    }
    std::lock_guard<std::mutex> one(run_mu_);
    {
      std::lock_guard<std::mutex> lk(mu_);
      fn_ = &fn;
      n_ = n;
      next_.store(0, std::memory_order_relaxed);
      active_ = (int)th_.size();
      ++gen_;
    }
    cv_.notify_all();
    drain(fn, n);
    std::unique_lock<std::mutex> lk(mu_);
    done_.wait(lk, [this] { return active_ == 0; });
    fn_ = nullptr;
  }

 private:
  void drain(const std::function<void(size_t)>& fn, size_t n) {
    for (;;) {
      const size_t i = next_.fetch_add(1, std::memory_order_relaxed);
      if (i >= n) return;
      fn(i);
    }
  }
  void loop() {
    uint64_t seen = 0;
    for (;;) {
      const std::function<void(size_t)>* fn;
      size_t n;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
        if (stop_) return;
        seen = gen_;
        fn = fn_;
        n = n_;
      }
      drain(*fn, n);
      {
        std::lock_guard<std::mutex> lk(mu_);
        if (--active_ == 0) done_.notify_one();
      }
    }
  }
  std::vector<std::thread> th_;
  std::mutex mu_, run_mu_;
  std::condition_variable cv_, done_;
  const std::function<void(size_t)>* fn_ = nullptr;
  size_t n_ = 0;
  std::atomic<size_t> next_{0};
  int active_ = 0;
  uint64_t gen_ = 0;
  bool stop_ = false;
};

// decimal digits of a non-negative int64 into `dst`; returns the number of characters written
static inline int put_i64(char* dst, int64_t v) {
  char tmp[24];
  int n = 0;
  if (v == 0) tmp[n++] = '0';
  while (v > 0) {
    tmp[n++] = (char)('0' + v % 10);
    v /= 10;
  }
  for (int i = 0; i < n; ++i) dst[i] = tmp[n - 1 - i];
  return n;
}

struct BulkError {
  int64_t index = -1;  // cut the error is about
  int64_t a = 0, b = 0;
  const char* what = nullptr;
};

// line i = head_i + mid[file_i] + "<byte offset>:<rows>:<cols>[:f16]" + tail_i + '\n'.
// Returns the bytes written, or -1 (err filled in): frame-count mismatch, or `cap` too small (err.a = bytes needed).
static int64_t manifest_lines(const char* heads, const int64_t* head_off, const char* tails, const int64_t* tail_off, int64_t batch,
                              const int64_t* frames, const int64_t* expected, const char* mids, const int64_t* mid_off, int32_t num_files,
                              const int32_t* file_of, const int64_t* byte_off, int32_t cols, int32_t bytes_per_value, char* out, int64_t cap,
                              BulkError* err) {
  if (expected)
    for (int64_t i = 0; i < batch; ++i)
      if (expected[i] >= 0 && expected[i] != frames[i]) {
        *err = BulkError{i, frames[i], expected[i], "frame-count contract"};
        return -1;
      }
  int64_t need = 0, mid_max = 0;
  for (int32_t k = 0; k < num_files; ++k) mid_max = std::max(mid_max, mid_off[k + 1] - mid_off[k]);
  for (int64_t i = 0; i < batch; ++i) {
    const int32_t k = file_of ? file_of[i] : 0;
    if (k < 0 || k >= num_files) {
      *err = BulkError{i, k, num_files, "file index out of range"};
      return -1;
    }
    need += (head_off[i + 1] - head_off[i]) + (tail_off[i + 1] - tail_off[i]) + mid_max + 64 + 1;
  }
  if (need > cap) {
    *err = BulkError{-1, need, cap, "output buffer too small"};
    return -1;
  }
  char* p = out;
  for (int64_t i = 0; i < batch; ++i) {
    const int32_t k = file_of ? file_of[i] : 0;
    const int64_t hl = head_off[i + 1] - head_off[i], tl = tail_off[i + 1] - tail_off[i], ml = mid_off[k + 1] - mid_off[k];
    std::memcpy(p, heads + head_off[i], (size_t)hl);
    p += hl;
    std::memcpy(p, mids + mid_off[k], (size_t)ml);
    p += ml;
    p += put_i64(p, byte_off[i]);
    *p++ = ':';
    p += put_i64(p, frames[i]);
    *p++ = ':';
    p += put_i64(p, cols);
    if (bytes_per_value == 2) {
      std::memcpy(p, ":f16", 4);
      p += 4;
    }
    std::memcpy(p, tails + tail_off[i], (size_t)tl);
    p += tl;
    *p++ = '\n';
  }
  return p - out;
}

// write all of [data, data + bytes) at `offset` of fd; 0 or errno
static int pwrite_all(int fd, const char* data, int64_t bytes, int64_t offset) {
  while (bytes > 0) {
    const ssize_t w = ::pwrite(fd, data, (size_t)std::min<int64_t>(bytes, (int64_t)1 << 30), (off_t)offset);
    if (w < 0) {
      if (errno == EINTR) continue;
      return errno;
    }
    data += w;
    offset += w;
    bytes -= w;
  }
  return 0;
}

// An append-only archive striped over a few files (the caller names them): a batch is cut into nfiles consecutive runs of whole cuts
// of about equal bytes and run k is appended to file k by its own thread.  One file = one inode = one set of page-cache locks: writers
// to DIFFERENT files do not serialise on them, which is what bounds a single tmpfs / page-cache file to one writer's copy rate
// (tools/tmpfs_write_probe.py).  With one file it is one pwrite on the calling thread.
constexpr int kErrNotFinite16 = -16;  // append(): a binary16 value of the batch is inf / nan

struct ArchiveFiles {
  std::vector<int> fds;
  std::vector<int64_t> size;  // bytes in file k
  WorkPool* pool = nullptr;   // nfiles - 1 persistent writer threads (the caller writes run 0)

  // -> 0, the errno of the first failing write (errfile = its index), or kErrNotFinite16.  file_of / byte_off: where every cut's rows
  // went.  check_f16: the rows are binary16 and must be finite (|x| <= 65504): every writer scans its own run before writing it.
  int append(const char* data, int64_t batch, const int64_t* frames, int64_t row_bytes, bool check_f16, int32_t* file_of, int64_t* byte_off, int* errfile) {
    const int nf = (int)fds.size();
    int64_t total = 0;
    for (int64_t i = 0; i < batch; ++i) total += frames[i] * row_bytes;
    // runs of whole cuts: cut i goes to file k while the bytes in front of it are below (k + 1) / nf of the batch
    std::vector<int64_t> start_byte(nf + 1, total);
    int64_t acc = 0;
    int k = 0;
    start_byte[0] = 0;
    for (int64_t i = 0; i < batch; ++i) {
      while (k + 1 < nf && acc * nf >= total * (int64_t)(k + 1)) {
        ++k;
        start_byte[k] = acc;
      }
      if (file_of) file_of[i] = k;
      if (byte_off) byte_off[i] = size[k] + (acc - start_byte[k]);
      acc += frames[i] * row_bytes;
    }
    for (int j = k + 1; j < nf; ++j) start_byte[j] = total;  // (fewer cuts than files: the remaining runs are empty)
    std::vector<int> rc(nf, 0);
    if (check_f16) {  // nothing is written unless the whole batch is finite
      auto scan = [&](size_t f) {
        const uint16_t* v = reinterpret_cast<const uint16_t*>(data + start_byte[f]);
        const int64_t n = (start_byte[f + 1] - start_byte[f]) / 2;
        unsigned worst = 0;
        for (int64_t i = 0; i < n; ++i) worst = std::max<unsigned>(worst, v[i] & 0x7FFFu);
        if (worst >= 0x7C00u) rc[f] = kErrNotFinite16;
      };
      if (pool) pool->run((size_t)nf, scan);
      else for (int f = 0; f < nf; ++f) scan((size_t)f);
      for (int f = 0; f < nf; ++f)
        if (rc[f]) {
          *errfile = f;
          return rc[f];
        }
    }
    auto write = [&](size_t f) {
      const int64_t n = start_byte[f + 1] - start_byte[f];
      if (n > 0) rc[f] = pwrite_all(fds[f], data + start_byte[f], n, size[f]);
    };
    if (pool) pool->run((size_t)nf, write);
    else for (int f = 0; f < nf; ++f) write((size_t)f);
    for (int f = 0; f < nf; ++f) {
      if (rc[f]) {
        // a failed append (ENOSPC on one stripe) leaves NO trace: every file is cut back to its size before the batch and no size[] moves,
        // so that rows never exist without manifest lines and a resumed run appends where the last complete batch ended
        for (int q = 0; q < nf; ++q) (void)::ftruncate(fds[q], (off_t)size[q]);
        *errfile = f;
        return rc[f];
      }
    }
    for (int f = 0; f < nf; ++f) size[f] += start_byte[f + 1] - start_byte[f];
    return 0;
  }
};

}  // namespace hipfeat
