// Host-side pieces of the bulk save path (lhotse/cut/set.py:2307-2363, lhotse/features/io.py:499-525): what the
// reference's _save_worker does per CUT in the interpreter -- store one matrix, build a Features object, validate its frame count,
// fastcopy the cut, to_dict, json.dumps -- restated per BATCH in plain C++, callable with the GIL released:
//   * manifest_lines: one JSONL line per cut = the two halves of the cut's own serialisation (made where the cut was loaded: the
//     loader's worker processes) spliced around the ONE field only the save path knows, the storage key, after the frame-count
//     contract of validate_features (lhotse/qa.py:286-301) has been checked for the whole batch;
//   * archive_append: the packed (sum T_b, F) matrix of a batch appended to the flat archive file(s) by a few writer threads.
// No HIP in this file.
#pragma once

#include <algorithm>
#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <unistd.h>

namespace hipfeat {

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
struct ArchiveFiles {
  std::vector<int> fds;
  std::vector<int64_t> size;  // bytes in file k

  // -> 0, or the errno of the first failing write (errfile = its index).  file_of / byte_off: where every cut's rows went.
  int append(const char* data, int64_t batch, const int64_t* frames, int64_t row_bytes, int32_t* file_of, int64_t* byte_off, int* errfile) {
    const int nf = (int)fds.size();
    int64_t total = 0;
    for (int64_t i = 0; i < batch; ++i) total += frames[i] * row_bytes;
    // runs of whole cuts: cut i goes to file k while the bytes in front of it are below (k + 1) / nf of the batch
    std::vector<int64_t> first(nf + 1, batch), start_byte(nf + 1, total);
    int64_t acc = 0;
    int k = 0;
    first[0] = 0;
    start_byte[0] = 0;
    for (int64_t i = 0; i < batch; ++i) {
      while (k + 1 < nf && acc * nf >= total * (int64_t)(k + 1)) {
        ++k;
        first[k] = i;
        start_byte[k] = acc;
      }
      if (file_of) file_of[i] = k;
      if (byte_off) byte_off[i] = size[k] + (acc - start_byte[k]);
      acc += frames[i] * row_bytes;
    }
    for (int j = k + 1; j < nf; ++j) {  // (fewer cuts than files: the remaining runs are empty)
      first[j] = batch;
      start_byte[j] = total;
    }
    std::vector<int> rc(nf, 0);
    std::vector<std::thread> th;
    auto run = [&](int f) {
      const int64_t n = start_byte[f + 1] - start_byte[f];
      if (n > 0) rc[f] = pwrite_all(fds[f], data + start_byte[f], n, size[f]);
    };
    for (int f = 1; f < nf; ++f)
      if (start_byte[f + 1] > start_byte[f]) th.emplace_back(run, f);
    run(0);
    for (auto& t : th) t.join();
    for (int f = 0; f < nf; ++f) {
      if (rc[f]) {
        *errfile = f;
        return rc[f];
      }
      size[f] += start_byte[f + 1] - start_byte[f];
    }
    return 0;
  }
};

}  // namespace hipfeat
