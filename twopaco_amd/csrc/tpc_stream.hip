// tpc_stream.hip -- the junction stream (the bytes of de_bruijn.bin) built on the device.
//
// Replaces, for the output pass, EdgeConstructionWorker's result vector + FlushEdgeResults
// (reference src/graphconstructor/vertexenumerator.h:837-854, 927-958) and
// JunctionPositionWriter::WriteJunction (src/common/junctionapi.h:118-132):
//   * every marked position with a real id becomes a 12-byte little-endian record (u32 position inside
//     its sequence, i64 id), in (sequence, position) order;
//   * the first and the last k-mer of every sequence of >= k bases that has no junction id gets a stub id
//     verticesCount + 42, + 43, ... in that order (VE.h:419, 942-948);
//   * before the first record of sequence c one separator (0xFFFFFFFF, INT64_MAX) per sequence-id step
//     (junctionapi.h:120-123); sequences shorter than k emit nothing but still consume an id.
// Slot arithmetic (slot = 12 bytes).  With n_r = records of sequence r (0 when shorter than k) and
// E_r = sum of n_r' over r' < r: the records of r start at slot E_r + r, and the separator that steps the
// writer from id j to j + 1 sits at slot E_(j+1) + j, for every j below the last emitting sequence.
#include "tpc_device.h"
#include "tpc_internal.h"
#include <rocprim/rocprim.hpp>
#include <algorithm>

namespace {

constexpr int64_t STREAM_INVALID = INT64_MAX;  // INVALID_VERTEX, common.cpp:5

__device__ __forceinline__ void put_record(uint32_t *out, uint64_t slot, uint32_t pos, int64_t id)
{
    uint32_t *p = out + slot * 3;
    p[0] = pos;
    p[1] = (uint32_t)((uint64_t)id & 0xFFFFFFFFull);
    p[2] = (uint32_t)((uint64_t)id >> 32);
}

__global__ void k_stream_flags(const int64_t *__restrict__ ids, uint64_t n, uint64_t *__restrict__ flags)
{   // n + 1 entries: the scan's last element is the total
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += stride) flags[i] = (i < n && ids[i] != STREAM_INVALID) ? 1ull : 0ull;
}

__device__ __forceinline__ uint64_t lower_bound_u64(const uint64_t *a, uint64_t n, uint64_t v)
{
    uint64_t lo = 0, hi = n;
    while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (a[mid] < v) lo = mid + 1; else hi = mid; }
    return lo;
}

struct RecPlan {
    uint64_t lo;        // first mark of the sequence
    uint32_t flags;     // bit 0: first k-mer has a real id, bit 1: last k-mer has one, bit 2: emits
};

// one thread per sequence: its marks, whether its end k-mers carry real ids, how many records and stubs it emits
__global__ void k_stream_plan(const uint64_t *__restrict__ rec_start, const uint64_t *__restrict__ rec_len, uint32_t n_rec, int k,
                              const uint64_t *__restrict__ marks, const int64_t *__restrict__ ids, const uint64_t *__restrict__ vscan, uint64_t n_marks,
                              RecPlan *__restrict__ plan, uint64_t *__restrict__ n_out, uint64_t *__restrict__ n_stub)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n_rec) return;
    if (r == n_rec) { n_out[r] = 0; n_stub[r] = 0; return; }  // scan tail
    const uint64_t len = rec_len[r];
    if (len < (uint64_t)k) { plan[r] = RecPlan{0, 0}; n_out[r] = 0; n_stub[r] = 0; return; }
    const uint64_t first = rec_start[r], last = first + len - k;
    const uint64_t lo = lower_bound_u64(marks, n_marks, first);
    const uint64_t hi = lower_bound_u64(marks, n_marks, last + 1);
    const bool has_first = lo < hi && marks[lo] == first && ids[lo] != STREAM_INVALID;
    const bool has_last = lo < hi && marks[hi - 1] == last && ids[hi - 1] != STREAM_INVALID;
    const uint32_t stubs = (has_first ? 0u : 1u) + ((last != first && !has_last) ? 1u : 0u);
    plan[r] = RecPlan{lo, (has_first ? 1u : 0u) | (has_last ? 2u : 0u) | 4u};
    n_out[r] = (uint64_t)(vscan[hi] - vscan[lo]) + stubs;
    n_stub[r] = stubs;
}

// one thread per sequence: separator after it, its stub records
__global__ void k_stream_fixed(const uint64_t *__restrict__ rec_start, const uint64_t *__restrict__ rec_len, uint32_t n_rec, int k, uint32_t r_last,
                               const RecPlan *__restrict__ plan, const uint64_t *__restrict__ e_scan, const uint64_t *__restrict__ s_scan,
                               uint64_t first_stub, uint32_t *__restrict__ out)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rec) return;
    if (r < r_last) put_record(out, e_scan[r + 1] + r, 0xFFFFFFFFu, STREAM_INVALID);
    const RecPlan p = plan[r];
    if (!(p.flags & 4u)) return;
    const uint64_t base = e_scan[r] + r, n = e_scan[r + 1] - e_scan[r];
    uint64_t stub = first_stub + s_scan[r];
    if (!(p.flags & 1u)) put_record(out, base, 0u, (int64_t)stub++);
    const uint64_t len = rec_len[r];
    if (len != (uint64_t)k && !(p.flags & 2u)) put_record(out, base + n - 1, (uint32_t)(len - k), (int64_t)stub);
}

// one thread per marked position with a real id
__global__ void k_stream_marks(const uint64_t *__restrict__ rec_start, uint32_t n_rec, const uint64_t *__restrict__ marks,
                               const int64_t *__restrict__ ids, const uint64_t *__restrict__ vscan, uint64_t n_marks, const RecPlan *__restrict__ plan,
                               const uint64_t *__restrict__ e_scan, uint32_t *__restrict__ out)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_marks; i += stride) {
        const int64_t id = ids[i];
        if (id == STREAM_INVALID) continue;
        const uint64_t g = marks[i];
        const uint32_t r = (uint32_t)(lower_bound_u64(rec_start, n_rec, g + 1) - 1);  // last sequence starting at or before g
        const RecPlan p = plan[r];
        const uint64_t slot = e_scan[r] + r + ((p.flags & 1u) ? 0u : 1u) + (vscan[i] - vscan[p.lo]);
        put_record(out, slot, (uint32_t)(g - rec_start[r]), id);
    }
}

// ---- the same stream cut over several ranks by text position ---------------------------------------------------
// A multi-GPU run leaves every rank with the marks (and ids) of its own chunk of the text [chunk_lo, chunk_hi).  Every slot
// of the stream belongs to one text position -- a record to its k-mer's position, the stub of a sequence's first / last
// k-mer to that k-mer's position, the separator that steps the writer from sequence j to j + 1 to the separator character in
// front of sequence j + 1 -- and slots are ordered as those positions are, so the slots of a chunk are one contiguous range of
// the file.  k_stream_partial gives what a rank knows about every sequence (real-id records among ITS marks, whether the end
// k-mers it holds carry real ids); the host layer adds the ranks up; k_stream_*_part then write exactly the rank's slots,
// relative to its first one.

// one thread per sequence: real-id records among this rank's marks, flags bit 0 / 1: this rank holds the first / last k-mer WITH a real id
__global__ void k_stream_partial(const uint64_t *__restrict__ rec_start, const uint64_t *__restrict__ rec_len, uint32_t n_rec, int k,
                                 const uint64_t *__restrict__ marks, const int64_t *__restrict__ ids, const uint64_t *__restrict__ vscan, uint64_t n_marks,
                                 uint64_t *__restrict__ cnt, uint32_t *__restrict__ flags, uint64_t *__restrict__ mark_lo)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rec) return;
    const uint64_t len = rec_len[r];
    if (len < (uint64_t)k) { cnt[r] = 0; flags[r] = 0; mark_lo[r] = 0; return; }
    const uint64_t first = rec_start[r], last = first + len - k;
    const uint64_t lo = lower_bound_u64(marks, n_marks, first);
    const uint64_t hi = lower_bound_u64(marks, n_marks, last + 1);
    const bool has_first = lo < hi && marks[lo] == first && ids[lo] != STREAM_INVALID;
    const bool has_last = lo < hi && marks[hi - 1] == last && ids[hi - 1] != STREAM_INVALID;
    cnt[r] = vscan[hi] - vscan[lo];
    flags[r] = (has_first ? 1u : 0u) | (has_last ? 2u : 0u);
    mark_lo[r] = lo;
}

// one thread per sequence: the separator and the stubs that belong to this rank's chunk.  gflags: bit 0 / 1 = the first / last
// k-mer has a real id on SOME rank, bit 2 = the sequence emits (>= k bases); e_scan / s_scan: records / stubs of the sequences before.
__global__ void k_stream_fixed_part(const uint64_t *__restrict__ rec_start, const uint64_t *__restrict__ rec_len, uint32_t n_rec, int k, uint32_t r_last,
                                    const uint32_t *__restrict__ gflags, const uint64_t *__restrict__ e_scan, const uint64_t *__restrict__ s_scan,
                                    uint64_t first_stub, uint64_t chunk_lo, uint64_t chunk_hi, uint64_t slot0, uint32_t *__restrict__ out)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rec) return;
    // separator r -> r + 1 belongs to the character in front of sequence r + 1
    if (r < r_last) {
        const uint64_t at = rec_start[r + 1] - 1;
        if (at >= chunk_lo && at < chunk_hi) put_record(out, e_scan[r + 1] + r - slot0, 0xFFFFFFFFu, STREAM_INVALID);
    }
    const uint32_t f = gflags[r];
    if (!(f & 4u)) return;
    const uint64_t base = e_scan[r] + r, n = e_scan[r + 1] - e_scan[r];
    const uint64_t len = rec_len[r], first = rec_start[r], last = first + len - k;
    uint64_t stub = first_stub + s_scan[r];
    if (!(f & 1u)) {
        if (first >= chunk_lo && first < chunk_hi) put_record(out, base - slot0, 0u, (int64_t)stub);
        stub++;
    }
    if (len != (uint64_t)k && !(f & 2u) && last >= chunk_lo && last < chunk_hi) put_record(out, base + n - 1 - slot0, (uint32_t)(len - k), (int64_t)stub);
}

// one thread per marked position of this rank with a real id.  before[r]: real-id records of sequence r on the ranks before this one
__global__ void k_stream_marks_part(const uint64_t *__restrict__ rec_start, uint32_t n_rec, const uint64_t *__restrict__ marks,
                                    const int64_t *__restrict__ ids, const uint64_t *__restrict__ vscan, uint64_t n_marks, const uint64_t *__restrict__ mark_lo,
                                    const uint32_t *__restrict__ gflags, const uint64_t *__restrict__ e_scan, const uint64_t *__restrict__ before, uint64_t slot0,
                                    uint32_t *__restrict__ out)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_marks; i += stride) {
        const int64_t id = ids[i];
        if (id == STREAM_INVALID) continue;
        const uint64_t g = marks[i];
        const uint32_t r = (uint32_t)(lower_bound_u64(rec_start, n_rec, g + 1) - 1);  // last sequence starting at or before g
        const uint64_t slot = e_scan[r] + r + ((gflags[r] & 1u) ? 0u : 1u) + before[r] + (vscan[i] - vscan[mark_lo[r]]);
        put_record(out, slot - slot0, (uint32_t)(g - rec_start[r]), id);
    }
}

template <class T>
int excl_scan(hipStream_t s, T *data, uint64_t n, void *&tmp, size_t &tmp_cap)
{
    size_t need = 0;
    if (rocprim::exclusive_scan(nullptr, need, data, data, T(0), n, rocprim::plus<T>(), s) != hipSuccess) return -2;
    if (need > tmp_cap) {
        if (tmp) (void)hipFree(tmp);
        tmp = nullptr; tmp_cap = 0;
        if (hipMalloc(&tmp, need) != hipSuccess) return -3;
        tmp_cap = need;
    }
    return rocprim::exclusive_scan(tmp, need, data, data, T(0), n, rocprim::plus<T>(), s) == hipSuccess ? 0 : -2;
}

}  // namespace

size_t tpc_stream_plan_bytes(uint32_t n_rec) { return ((size_t)n_rec + 1) * (sizeof(RecPlan) + 2 * sizeof(uint64_t)) + 64; }

// scratch: vscan (n_marks + 1 uint64), rec (tpc_stream_plan_bytes(n_rec)); d_rec_start / d_rec_len on the device.
// totals_host[0] = records (junction occurrences + stubs), totals_host[1] = slots (records + separators).
int tpc_launch_stream_plan(hipStream_t s, const uint64_t *d_rec_start, const uint64_t *d_rec_len, uint32_t n_rec, int k, const uint64_t *marks,
                           const int64_t *ids, uint64_t n_marks, uint64_t *vscan, void *rec, uint32_t r_last, uint64_t *totals_host)
{
    void *tmp = nullptr;
    size_t tmp_cap = 0;
    int rc = 0;
    RecPlan *plan = (RecPlan *)rec;
    uint64_t *e_scan = (uint64_t *)(plan + n_rec + 1), *s_scan = e_scan + n_rec + 1;
    hipLaunchKernelGGL(k_stream_flags, dim3((unsigned)std::min<uint64_t>((n_marks + 256) / 256, 4096)), dim3(256), 0, s, ids, n_marks, vscan);
    if ((rc = excl_scan<uint64_t>(s, vscan, n_marks + 1, tmp, tmp_cap)) == 0) {
        hipLaunchKernelGGL(k_stream_plan, dim3((n_rec + 256) / 256), dim3(256), 0, s, d_rec_start, d_rec_len, n_rec, k, marks, ids, vscan, n_marks, plan,
                           e_scan, s_scan);
        if ((rc = excl_scan<uint64_t>(s, e_scan, (uint64_t)n_rec + 1, tmp, tmp_cap)) == 0) rc = excl_scan<uint64_t>(s, s_scan, (uint64_t)n_rec + 1, tmp, tmp_cap);
    }
    uint64_t records = 0;
    if (rc == 0 && hipMemcpyAsync(&records, e_scan + n_rec, sizeof records, hipMemcpyDeviceToHost, s) != hipSuccess) rc = -2;
    if (hipStreamSynchronize(s) != hipSuccess) rc = rc ? rc : -2;
    if (tmp) (void)hipFree(tmp);
    if (rc) return rc;
    totals_host[0] = records;
    totals_host[1] = records + (records ? r_last : 0);
    return 0;
}

int tpc_launch_stream_write(hipStream_t s, const uint64_t *d_rec_start, const uint64_t *d_rec_len, uint32_t n_rec, int k, const uint64_t *marks,
                            const int64_t *ids, uint64_t n_marks, const uint64_t *vscan, const void *rec, uint32_t r_last, uint64_t first_stub, uint32_t *out)
{
    const RecPlan *plan = (const RecPlan *)rec;
    const uint64_t *e_scan = (const uint64_t *)(plan + n_rec + 1), *s_scan = e_scan + n_rec + 1;
    hipLaunchKernelGGL(k_stream_fixed, dim3((n_rec + 255) / 256), dim3(256), 0, s, d_rec_start, d_rec_len, n_rec, k, r_last, plan, e_scan, s_scan, first_stub, out);
    if (n_marks)
        hipLaunchKernelGGL(k_stream_marks, dim3((unsigned)std::min<uint64_t>((n_marks + 255) / 256, 8192)), dim3(256), 0, s, d_rec_start, n_rec, marks, ids,
                           vscan, n_marks, plan, e_scan, out);
    return 0;
}

// sharded stream, step 1: vscan (n_marks + 1), then per-sequence partial counts / flags / first mark (device arrays of n_rec)
int tpc_launch_stream_partial(hipStream_t s, const uint64_t *d_rec_start, const uint64_t *d_rec_len, uint32_t n_rec, int k, const uint64_t *marks,
                              const int64_t *ids, uint64_t n_marks, uint64_t *vscan, uint64_t *cnt, uint32_t *flags, uint64_t *mark_lo)
{
    void *tmp = nullptr;
    size_t tmp_cap = 0;
    hipLaunchKernelGGL(k_stream_flags, dim3((unsigned)std::min<uint64_t>((n_marks + 256) / 256, 4096)), dim3(256), 0, s, ids, n_marks, vscan);
    int rc = excl_scan<uint64_t>(s, vscan, n_marks + 1, tmp, tmp_cap);
    if (rc == 0) hipLaunchKernelGGL(k_stream_partial, dim3((n_rec + 255) / 256), dim3(256), 0, s, d_rec_start, d_rec_len, n_rec, k, marks, ids, vscan, n_marks, cnt, flags, mark_lo);
    if (hipStreamSynchronize(s) != hipSuccess) rc = rc ? rc : -2;
    if (tmp) (void)hipFree(tmp);
    return rc;
}

// sharded stream, step 2: this rank's slots [slot0, slot0 + n_slots) into `out` (relative to slot0)
int tpc_launch_stream_write_part(hipStream_t s, const uint64_t *d_rec_start, const uint64_t *d_rec_len, uint32_t n_rec, int k, const uint64_t *marks,
                                 const int64_t *ids, uint64_t n_marks, const uint64_t *vscan, const uint64_t *mark_lo, const uint32_t *gflags,
                                 const uint64_t *e_scan, const uint64_t *s_scan, const uint64_t *before, uint32_t r_last, uint64_t first_stub,
                                 uint64_t chunk_lo, uint64_t chunk_hi, uint64_t slot0, uint32_t *out)
{
    hipLaunchKernelGGL(k_stream_fixed_part, dim3((n_rec + 255) / 256), dim3(256), 0, s, d_rec_start, d_rec_len, n_rec, k, r_last, gflags, e_scan, s_scan, first_stub,
                       chunk_lo, chunk_hi, slot0, out);
    if (n_marks)
        hipLaunchKernelGGL(k_stream_marks_part, dim3((unsigned)std::min<uint64_t>((n_marks + 255) / 256, 8192)), dim3(256), 0, s, d_rec_start, n_rec, marks, ids,
                           vscan, n_marks, mark_lo, gflags, e_scan, before, slot0, out);
    return 0;
}

// tpc_preload: the first use of any kernel of this translation unit makes the runtime load its code object
__global__ void k_warm_stream() {}
int tpc_warm_stream() { hipFuncAttributes a; return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(k_warm_stream)) == hipSuccess ? 0 : -1; }
