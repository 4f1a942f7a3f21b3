// t4d_raster_sort.h - part of the translation unit t4d_raster.hip (included there, inside its anonymous namespace; not a
// stand-alone header).  A.2: per-tile sort by (depth bits, Gaussian index): register bitonic runs, ranking merges, long-bin sort.
// See t4d_raster.hip for the overview, the constants, the state layout and the kernel parameter block.
// ---------------------------------------------------------------------------------------------------------
// A.2 per-tile sort by (depth bits, Gaussian index)
// ---------------------------------------------------------------------------------------------------------
// Sort the 64 keys of a wave (one per lane) ascending, entirely in registers: bitonic network whose exchanges are DPP
// moves (xor 1, 2: quad_perm; xor 4: two bank-masked row shifts; xor 8: row_ror:8) or ds_bpermute (xor 16, 32).
template <int J>
__device__ __forceinline__ uint32_t lane_xor(const uint32_t v)
{
    if (J == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);
    if (J == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true);
    if (J == 4) {
        const int t = __builtin_amdgcn_update_dpp(0, (int)v, 0x104, 0xf, 0x5, true);      // banks {0,2} read lane+4
        return (uint32_t)__builtin_amdgcn_update_dpp(t, (int)v, 0x114, 0xf, 0xA, true);  // banks {1,3} read lane-4
    }
    if (J == 8) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, true);
    return (uint32_t)__shfl_xor((int)v, J, 64);
}

template <int K, int J>
__device__ __forceinline__ void bitonic_step(unsigned long long &key, const int lane)
{
    const unsigned long long other = ((unsigned long long)lane_xor<J>((uint32_t)(key >> 32)) << 32) | lane_xor<J>((uint32_t)key);
    const bool keep_min = ((lane & J) == 0) == ((lane & K) == 0);      // K = 64: every lane sorts ascending
    key = ((other < key) == keep_min) ? other : key;
    if constexpr (J > 1) bitonic_step<K, J / 2>(key, lane);
}

__device__ __forceinline__ void wave_sort64(unsigned long long &key, const int lane)
{
    bitonic_step<2, 1>(key, lane);
    bitonic_step<4, 2>(key, lane);
    bitonic_step<8, 4>(key, lane);
    bitonic_step<16, 8>(key, lane);
    bitonic_step<32, 16>(key, lane);
    bitonic_step<64, 32>(key, lane);
}

// number of keys smaller than `key` in a sorted run of 64 (branch-free binary search, 7 LDS reads)
__device__ __forceinline__ uint32_t run_lower_bound(const unsigned long long *run, const unsigned long long key)
{
    uint32_t pos = 0;
#pragma unroll
    for (int st = 32; st > 0; st >>= 1)
        if (run[pos + st - 1] < key) pos += st;
    return pos + (run[pos] < key ? 1u : 0u);
}

// number of keys smaller than `key` among run[0..len) (sorted, global memory)
__device__ __forceinline__ uint32_t lower_bound_global(const unsigned long long *run, const uint32_t len, const uint32_t cap2,
                                                       const unsigned long long key)
{
    uint32_t pos = 0;
    for (uint32_t st = cap2 >> 1; st > 0; st >>= 1)                  // cap2 = power of two >= len
        if (pos + st <= len && run[pos + st - 1] < key) pos += st;
    return pos + ((pos < len && run[pos] < key) ? 1u : 0u);
}

// Sort n <= kSortLdsCap keys (global memory, in place) through the workgroup's LDS buffer: runs of 64 are sorted in
// registers, then at every level each key finds its slot in the merged pair of runs as (position in its own run) + (keys of
// the sibling run below it), log2(width)+1 dependent LDS reads; keys wait in registers between the read and the write phase.
// log2(n/64) levels with two barriers each (a compare-exchange network needs ~60 barriers at this size).
// PRE != 0: the keys arrive as sorted runs of PRE (k_sort_long_chunks): only the levels from there on are left.
// dst: where the sorted keys go (default: in place)
template <int BLOCK, int CAP, int PRE = 0>
__device__ __forceinline__ void sort_chunk_lds(unsigned long long *keys, const uint32_t n, unsigned long long *s_keys,
                                               const int tid, const int wave, const int lane, unsigned long long *dst = nullptr)
{
    if (dst == nullptr) dst = keys;
    constexpr int kPer = CAP / BLOCK;
    const uint32_t runs = (n + 63u) >> 6, N = runs << 6;
    for (uint32_t r = (uint32_t)wave; r < runs; r += BLOCK / 64) {
        const uint32_t i = (r << 6) + (uint32_t)lane;
        unsigned long long k0 = i < n ? keys[i] : ~0ull;           // the last run is padded with +inf
        if (PRE == 0) wave_sort64(k0, lane);
        s_keys[i] = k0;
    }
    __syncthreads();
    for (uint32_t w = PRE != 0 ? (uint32_t)PRE : 64u; w < N; w <<= 1) {
        unsigned long long kk[kPer];
        uint32_t np[kPer];
#pragma unroll
        for (int e = 0; e < kPer; e++) {
            const uint32_t p = (uint32_t)tid + e * BLOCK;
            if (p < N) {
                kk[e] = s_keys[p];
                const uint32_t run = p / w, sbase = (run ^ 1u) * w;
                const uint32_t slen = sbase < N ? min(w, N - sbase) : 0u;
                const unsigned long long *sib = s_keys + sbase;
                uint32_t pos = 0;
                for (uint32_t st = w >> 1; st > 0; st >>= 1)
                    if (pos + st <= slen && sib[pos + st - 1] < kk[e]) pos += st;
                if (pos < slen && sib[pos] < kk[e]) pos++;
                np[e] = (run & ~1u) * w + (p & (w - 1u)) + pos;
            }
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < kPer; e++)
            if ((uint32_t)tid + e * BLOCK < N) s_keys[np[e]] = kk[e];
        __syncthreads();
    }
    for (uint32_t i = tid; i < n; i += BLOCK) dst[i] = s_keys[i];
    __syncthreads();
}

// Sort a bin longer than the LDS buffer: chunks of CAP keys are sorted through the LDS, then merged level by level IN GLOBAL
// MEMORY by the same ranking step, ping-pong between the key arena and the scratch arena of the same size (the bin's keys
// stay in this XCD's L2).  Four independent binary searches per thread and step overlap their latencies.
template <int BLOCK, int CAP, int PRE = 0>
__device__ __forceinline__ void sort_bin_chunked(unsigned long long *keys, unsigned long long *tmp, const uint32_t n,
                                                 unsigned long long *s_keys, const int tid, const int wave, const int lane)
{
    for (uint32_t c = 0; c < n; c += CAP) sort_chunk_lds<BLOCK, CAP, PRE>(keys + c, min((uint32_t)CAP, n - c), s_keys, tid, wave, lane);
    unsigned long long *src = keys, *dst = tmp;
    for (uint32_t w = CAP; w < n; w <<= 1) {
        __threadfence_block();
        __syncthreads();                                   // the previous level's writes are visible to the workgroup
        for (uint32_t i0 = (uint32_t)tid * 4u; i0 < n; i0 += BLOCK * 4u) {
            unsigned long long kk[4];
            uint32_t slot[4];
#pragma unroll
            for (int e = 0; e < 4; e++) kk[e] = i0 + e < n ? src[i0 + e] : ~0ull;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const uint32_t i = i0 + e, run = i / w, sbase = (run ^ 1u) * w;
                const uint32_t slen = sbase < n ? min(w, n - sbase) : 0u;
                slot[e] = (run & ~1u) * w + (i & (w - 1u)) + lower_bound_global(src + sbase, slen, w, kk[e]);
            }
#pragma unroll
            for (int e = 0; e < 4; e++)
                if (i0 + e < n) dst[slot[e]] = kk[e];
        }
        unsigned long long *t2 = src; src = dst; dst = t2;
    }
    if (src != keys) {
        __threadfence_block();
        __syncthreads();
        for (uint32_t i = tid; i < n; i += BLOCK) keys[i] = src[i];
    }
}

// One bin of n keys, sorted by the whole workgroup (BLOCK threads) through s_keys (kSortLdsCap keys).  KEEP: leave the sorted keys
// in s_keys[0, n) as well (n <= kSortLdsCap) - the latency build of k_render_fwd sorts its own tile's bin and stages from there.
template <bool KEEP, int BLOCK>
__device__ __forceinline__ void sort_one_bin(const KP &kp, const int v, const uint32_t off, const uint32_t n, unsigned long long *s_keys,
                                             const int tid, const int wave, const int lane)
{
    unsigned long long *keys = kp.keys + (size_t)v * kp.cap + off;
    if (n <= (uint32_t)kRankSortMax) {
        // Runs of 64 keys are sorted inside a wave's registers (no LDS, no barrier); a key's final position is its
        // position in its own run plus, for every other run, the number of keys smaller than it (keys are unique: the
        // Gaussian index is the low word).  One barrier per tile, 7 dependent LDS reads per (key, other run).
        constexpr int kWaves = BLOCK / 64;
        constexpr int kPer = kRankSortMax / BLOCK > 0 ? kRankSortMax / BLOCK : 1;      // keys per thread
        const uint32_t runs = (n + 63u) >> 6;
        unsigned long long mine[kPer];
        uint32_t ranks[kPer];
#pragma unroll
        for (int e = 0; e < kPer; e++) {
            const uint32_t i = (uint32_t)tid + e * BLOCK;      // run (wave + kWaves e), position lane
            if ((uint32_t)(wave + kWaves * e) < runs) {        // wave-uniform
                mine[e] = i < n ? keys[i] : ~0ull;             // the last run is padded with +inf
                wave_sort64(mine[e], lane);
                s_keys[i] = mine[e];
            }
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < kPer; e++) {
            const uint32_t own = (uint32_t)(wave + kWaves * e);
            ranks[e] = 0xffffffffu;
            if (own < runs) {
                uint32_t rank = (uint32_t)lane;
#pragma unroll
                for (uint32_t r = 0; r < (uint32_t)(kRankSortMax / 64); r++)      // unrolled: the searches overlap
                    if (r < runs && r != own) rank += run_lower_bound(s_keys + ((r % kWaves) * 64u + (r / kWaves) * BLOCK), mine[e]);
                if (mine[e] != ~0ull) { keys[rank] = mine[e]; ranks[e] = rank; }
            }
        }
        if (KEEP) {
            __syncthreads();                                   // every search has read the runs: they may be overwritten
#pragma unroll
            for (int e = 0; e < kPer; e++)
                if (ranks[e] != 0xffffffffu) s_keys[ranks[e]] = mine[e];
        }
    } else if (n <= (uint32_t)kSortLdsCap) {
        sort_chunk_lds<BLOCK, kSortLdsCap>(keys, n, s_keys, tid, wave, lane);
    } else if (!kp.long_bins_elsewhere) {
        // only when the host said that no such bin exists (T4D_FLAG_NO_LONG_BINS) and one appeared nevertheless:
        // correct, but one workgroup per bin with 16 KiB of LDS - k_sort_long is the fast path
        sort_bin_chunked<BLOCK, kSortLdsCap>(keys, kp.sort_tmp + (size_t)v * kp.cap + off, n, s_keys, tid, wave, lane);
    }
}

// BLOCK = 256: the throughput build (a 24-view launch holds thousands of bins: four waves per bin keep every SIMD busy).
// BLOCK = 1024: small launches (at most kSegMaxTiles tiles), whose sort lasts as long as its LONGEST bin takes one workgroup:
// a lone wave issues an instruction every four cycles, so a bin of 1,286 keys took 28 us on four waves (six register sorts of
// ~1 us and 5 merge levels of ~3 us per wave: tools/micro/sort_bin.hip); sixteen waves share that work.
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_sort_tiles(const KP kp)
{
    constexpr int kWaves = BLOCK / 64;
    __shared__ unsigned long long s_keys[kSortLdsCap];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // Work units.  The item list is ordered by length class (floor(log2 n), descending), and bucket_fill holds the size of every
    // class: bins of 64 keys and more are one unit per workgroup; bins of 2..63 keys fit one register-sorted run, need neither LDS
    // nor a barrier, and go one PER WAVE to a unit (a high-resolution pass has mostly such bins: config 4 averages 65 keys
    // per non-empty tile, and three of the four waves of a one-bin workgroup did nothing).
    constexpr int kClass63 = (kBuckets - 2) - 5, kClass1 = kBuckets - 2;       // classes of n in [32, 63] and of n == 1
    uint32_t n_big = 0, n_small = 0;
#pragma unroll
    for (int k = 0; k < kBuckets - 1; k++) {
        const uint32_t f = kp.bucket_fill[k];
        if (k < kClass63) n_big += f;
        else if (k < kClass1) n_small += f;
    }
    const uint32_t n_units = n_big + ((n_small + kWaves - 1u) / kWaves);
    for (uint32_t unit = blockIdx.x; unit < n_units; unit += gridDim.x) {
        if (unit >= n_big) {
            const uint32_t item = n_big + (uint32_t)kWaves * (unit - n_big) + (uint32_t)wave;     // wave-uniform
            if (item < n_big + n_small) {
                const uint4 it = kp.items[item];
                const uint32_t n = it.z;
                unsigned long long *keys = kp.keys + (size_t)(it.x >> 20) * kp.cap + it.y;
                unsigned long long k0 = (uint32_t)lane < n ? keys[lane] : ~0ull;
                wave_sort64(k0, lane);
                if ((uint32_t)lane < n) keys[lane] = k0;
            }
            continue;
        }
        const uint4 it = kp.items[unit];
        sort_one_bin<false, BLOCK>(kp, (int)(it.x >> 20), it.y, it.z, s_keys, tid, wave, lane);
        __syncthreads();                                           // s_keys is reused by the next item
    }
}

// Bins longer than kSortLdsCap keys (dense passes: 191 of 11,544 non-empty bins at P = 1M, 4096x3008, the longest 15,693 keys).
// Work items are ordered by length class, so the long bins come first and a workgroup stops at the first bin of a shorter class.
// Two launches, in both of which every kSortLdsCap-key CHUNK of every long bin is a work unit of its own, taken by whichever
// workgroup it falls to (a bin of 12,614 keys: seven units in parallel):
//   k_sort_long_chunks: sorts the chunk through 16 KiB of LDS and leaves it in the scratch arena (sort_tmp), at the bin's offsets;
//   k_merge_long: ONE ranking pass instead of log2(chunks) merge levels - a key's final position in its bin is its position in its
//     own chunk plus, for every other chunk, the number of keys there that are smaller (keys are unique: the Gaussian index is the
//     low word); two keys per thread, the binary searches of four sibling chunks in lock-step (eight independent chains of
//     dependent loads from this XCD's L2), keys scattered from the scratch arena to their final slots in the key arena.  Any bin
//     length: no LDS, no cap.
// (History, one 4096 x 3008 view of 10^6 Gaussians: as one kernel - runs of 64, then eight merge levels by one workgroup - the long
// bins took 93 us behind k_sort_tiles' 45; chunks in parallel + a whole CU per bin merging up to 16,384 keys in 128 KiB of LDS: 20 +
// 37-52 us, the launch lasting as long as its longest bin's three merge levels on ONE workgroup; the ranking pass with its searches
// in memory: 20 + 38 us; with the siblings staged in LDS: 20 + 17-21 us - round 6.)
constexpr int kLongScan = 1024;          // long bins a workgroup counts in one go

// Calls unit(work item, first key of the chunk) for this workgroup's share of the chunks of all long bins: units u = blockIdx.x,
// + gridDim.x, ... in the global numbering (bins in work-item order, chunks in order).  s_first [kLongScan + 1], s_wsum [kLongBlock / 64].
template <typename F>
__device__ __forceinline__ void for_each_long_chunk(const KP &kp, uint32_t *s_first, uint32_t *s_wsum, F &&unit)
{
    static_assert(kLongScan == kLongBlock, "one work item per thread and round");
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint32_t n_items = (uint32_t)(kp.V * kp.T);
    uint32_t units_before = 0;                           // units of the rounds before this one
    for (uint32_t base = 0; base < n_items; base += kLongScan) {
        // chunks of this round's work items (0 for a bin that k_sort_tiles sorts)
        const uint32_t i = base + (uint32_t)tid;
        const uint32_t n_i = i < n_items ? kp.items[i].z : 0u;
        const uint32_t c_i = n_i > (uint32_t)kSortLdsCap ? (n_i + (uint32_t)kSortLdsCap - 1u) / (uint32_t)kSortLdsCap : 0u;
        const uint32_t incl = wave_incl_scan(c_i);
        if (lane == 63) s_wsum[wave] = incl;
        // does the list go on with long bins behind this round?  (ordered by length class: not once a shorter class has begun)
        const bool more = __syncthreads_or(tid == kLongScan - 1 && n_i >= (uint32_t)kSortLdsCap);
        uint32_t woff = 0, total = 0;
#pragma unroll
        for (int w = 0; w < kLongBlock / 64; w++) {
            const uint32_t x = s_wsum[w];
            if (w < wave) woff += x;
            total += x;
        }
        s_first[tid] = woff + incl - c_i;
        if (tid == 0) s_first[kLongScan] = total;
        __syncthreads();
        // this workgroup's units of the round: u = blockIdx.x, + gridDim.x, ... (global unit numbers)
        for (uint32_t u = blockIdx.x + ((units_before + gridDim.x - 1u - blockIdx.x) / gridDim.x) * gridDim.x; u < units_before + total; u += gridDim.x) {
            const uint32_t lu = u - units_before;
            uint32_t pos = 0;                            // the work item that holds local unit lu: the last one with s_first <= lu
#pragma unroll
            for (uint32_t st = kLongScan >> 1; st > 0; st >>= 1)
                if (s_first[pos + st] <= lu) pos += st;
            unit(kp.items[base + pos], (lu - s_first[pos]) * (uint32_t)kSortLdsCap);
        }
        units_before += total;
        if (!more) break;
        __syncthreads();                                 // (s_first, s_wsum are rewritten by the next round)
    }
}

__global__ __launch_bounds__(kLongBlock) void k_sort_long_chunks(const KP kp)
{
    __shared__ unsigned long long s_keys[kSortLdsCap];
    __shared__ uint32_t s_first[kLongScan + 1];          // s_first[i] = units in front of work item i
    __shared__ uint32_t s_wsum[kLongBlock / 64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for_each_long_chunk(kp, s_first, s_wsum, [&](const uint4 it, const uint32_t c) {
        const size_t base = (size_t)(it.x >> 20) * kp.cap + it.y + c;
        sort_chunk_lds<kLongBlock, kSortLdsCap>(kp.keys + base, min((uint32_t)kSortLdsCap, it.z - c), s_keys, tid, wave, lane, kp.sort_tmp + base);
    });
}

__global__ __launch_bounds__(kLongBlock) void k_merge_long(const KP kp)
{
    constexpr int kGroup = 4;                            // sibling chunks staged in LDS at a time (64 KiB: two workgroups per CU)
    __shared__ unsigned long long s_sib[kGroup * kSortLdsCap];
    __shared__ uint32_t s_first[kLongScan + 1];
    __shared__ uint32_t s_wsum[kLongBlock / 64];
    constexpr int kPer = kSortLdsCap / kLongBlock;       // keys of a chunk per thread
    static_assert(kPer == 2, "two keys per thread, their searches in lock-step");
    const int tid = threadIdx.x;
    for_each_long_chunk(kp, s_first, s_wsum, [&](const uint4 it, const uint32_t c) {
        const uint32_t n = it.z;
        const size_t base = (size_t)(it.x >> 20) * kp.cap + it.y;
        const unsigned long long *src = kp.sort_tmp + base;
        unsigned long long *dst = kp.keys + base;
        const uint32_t len = min((uint32_t)kSortLdsCap, n - c);
        unsigned long long key[kPer];
        uint32_t rank[kPer];
#pragma unroll
        for (int e = 0; e < kPer; e++) {
            const uint32_t i = (uint32_t)tid + (uint32_t)e * kLongBlock;
            key[e] = i < len ? src[c + i] : ~0ull;
            rank[e] = i;                                 // position in the own chunk; every other chunk adds its keys below
        }
        // The sibling chunks come through LDS, kGroup at a time: a search is a chain of 12 dependent reads, and from memory each of
        // them is a trip to the Infinity Cache (the chunks were written by other XCDs a kernel ago: ~0.7 us per step measured, 38 us
        // for the launch); staged, a group costs one round of independent 16-byte loads and 12 LDS reads.
        for (uint32_t s0 = 0; s0 < n; s0 += (uint32_t)(kGroup * kSortLdsCap)) {
            const uint32_t glen = min((uint32_t)(kGroup * kSortLdsCap), n - s0);
            __syncthreads();                             // (the previous group's - or unit's - searches have left the buffer)
            for (uint32_t i = (uint32_t)tid * 2u; i < glen; i += 2u * kLongBlock) {
                // (base and s0 are even multiples of 8 bytes only by luck: 8-byte loads; two per thread and round keep them in flight)
                const unsigned long long a = src[s0 + i];
                const unsigned long long b = i + 1u < glen ? src[s0 + i + 1u] : ~0ull;
                s_sib[i] = a; s_sib[i + 1u] = b;
            }
            __syncthreads();
            uint32_t lb[kPer][kGroup], sl[kGroup];
#pragma unroll
            for (int j = 0; j < kGroup; j++) {
                const uint32_t s = s0 + (uint32_t)j * (uint32_t)kSortLdsCap;
                sl[j] = (s < n && s != c) ? min((uint32_t)kSortLdsCap, n - s) : 0u;
#pragma unroll
                for (int e = 0; e < kPer; e++) lb[e][j] = 0u;
            }
            for (uint32_t st = (uint32_t)kSortLdsCap >> 1; st > 0; st >>= 1) {
#pragma unroll
                for (int j = 0; j < kGroup; j++)
#pragma unroll
                    for (int e = 0; e < kPer; e++) {
                        const uint32_t p = lb[e][j] + st;
                        if (p <= sl[j] && s_sib[j * kSortLdsCap + p - 1u] < key[e]) lb[e][j] = p;
                    }
            }
#pragma unroll
            for (int j = 0; j < kGroup; j++)
#pragma unroll
                for (int e = 0; e < kPer; e++)
                    rank[e] += lb[e][j] + ((lb[e][j] < sl[j] && s_sib[j * kSortLdsCap + lb[e][j]] < key[e]) ? 1u : 0u);
        }
#pragma unroll
        for (int e = 0; e < kPer; e++)
            if ((uint32_t)tid + (uint32_t)e * kLongBlock < len) dst[rank[e]] = key[e];
    });
}
