// What limits the DiT linears' operand fetch: global -> LDS by global_load_lds (16 B per lane) of a GEMM workgroup's A and W tiles
// through a ring of stages, NO MFMAs - the ring kernel's staging traffic alone.  Compares row segments of 64 bytes per stage
// (BK = 32, what gemm256_k stages: two stages fetch the two halves of every 128-byte line separately) with 128 bytes (BK = 64:
// one request per line), at the tile shapes and workgroup counts of the products at M = 6400.
//   hipcc -O3 --offload-arch=gfx950 tests/micro/fetch_rows.hip -o tests/micro/fetch_rows_bench && tests/micro/fetch_rows_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__device__ int g_same;          // 1: every workgroup fetches tile (0, 0) - everything hits L2; 0: its own tile
template <int BK, int STAGES, int NWAVE>
__global__ __launch_bounds__(NWAVE * 64) void fetch_k(const unsigned short* __restrict__ A, const unsigned short* __restrict__ W, int M, int N, int K, int BM, int BN, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int ntn = N / BN, nwg = gridDim.x;
    const int orig = blockIdx.x, xcd = orig & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    const int m0 = g_same ? 0 : (wg / ntn) * BM, n0 = g_same ? 0 : (wg % ntn) * BN;
    constexpr int RB = BK * 2;                    // bytes per row per stage
    constexpr int LPR = RB / 16;                  // lanes per row
    constexpr int RPI = 64 / LPR;                 // rows per wave-instruction (1 KiB)
    const int rows = BM + BN, n_inst = rows / RPI;              // wave-instructions per stage
    const int stage_bytes = rows * RB;
    const int nt = K / BK;
    auto issue = [&](int t) {
        char* st = smem + (t % STAGES) * stage_bytes;
        for (int j = wid; j < n_inst; j += NWAVE) {
            const int row = j * RPI + lane / LPR, c = lane % LPR;
            const unsigned short* src = row < BM ? A + (long)min(m0 + row, M - 1) * K : W + (long)min(n0 + row - BM, N - 1) * K;
            __builtin_amdgcn_global_load_lds((const void*)(src + t * BK + c * 8), (__attribute__((address_space(3))) void*)(st + j * 1024), 16, 0, 0);
        }
    };
    for (int t = 0; t < STAGES - 1 && t < nt; ++t) issue(t);
    int acc = 0;
    for (int t = 0; t < nt; ++t) {
        // wait for everything but the stages issued after t (simplest correct form: all of this wave's older DMAs)
        if (t + STAGES - 1 < nt) issue(t + STAGES - 1);
        // the ring kernel waits with a counted vmcnt; here: the DMAs of stage t are the oldest, wait until only the younger remain
        const int per = (n_inst + NWAVE - 1 - wid) / NWAVE;              // this wave's DMAs per stage
        const int younger = min(nt - 1 - t, STAGES - 1) * per;
        if (younger >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (younger >= 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else if (younger >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (younger >= 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (younger >= 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else if (younger >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (younger >= 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else if (younger >= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else if (younger >= 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        acc += *reinterpret_cast<volatile int*>(smem + (t % STAGES) * stage_bytes + tid * 4);
        __builtin_amdgcn_s_barrier();            // the slot may be refilled
    }
    if (acc == 0x7fffffff) sink[0] = acc;
}

// The same tiles through REGISTERS: global_load_dwordx4 (16 B per lane, full 128-byte row segments at BK = 64) a stage ahead, then
// ds_write_b128 into a double-buffered LDS tile - what the library GEMMs do.  NL = 16-byte loads per thread and stage.
template <int BK, int NWAVE, int NL>
__global__ __launch_bounds__(NWAVE * 64) void fetch_reg_k(const unsigned short* __restrict__ A, const unsigned short* __restrict__ W, int M, int N, int K, int BM, int BN, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int ntn = N / BN, nwg = gridDim.x;
    const int orig = blockIdx.x, xcd = orig & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    const int m0 = (wg / ntn) * BM, n0 = (wg % ntn) * BN;
    constexpr int RB = BK * 2, LPR = RB / 16, NT = NWAVE * 64;
    const int rows = BM + BN, stage_bytes = rows * RB, nt = K / BK;
    const unsigned short* src[NL];
    int dst[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int idx = tid + i * NT, row = min(idx / LPR, rows - 1), c = idx % LPR;
        src[i] = (row < BM ? A + (long)min(m0 + row, M - 1) * K : W + (long)min(n0 + row - BM, N - 1) * K) + c * 8;
        dst[i] = row * RB + ((c ^ (row & (LPR - 1))) << 4);
    }
    uint4 r[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) r[i] = *reinterpret_cast<const uint4*>(src[i]);
    int acc = 0;
    for (int t = 0; t < nt; ++t) {
        char* st = smem + (t & 1) * stage_bytes;
#pragma unroll
        for (int i = 0; i < NL; ++i) *reinterpret_cast<uint4*>(st + dst[i]) = r[i];
        if (t + 1 < nt) {
#pragma unroll
            for (int i = 0; i < NL; ++i) r[i] = *reinterpret_cast<const uint4*>(src[i] + (t + 1) * BK);
        }
        __syncthreads();
        acc += *reinterpret_cast<volatile int*>(st + tid * 4);
    }
    if (acc == 0x7fffffff) sink[0] = acc;
}

// Pure L2 -> register rate for the same row segments: NWAVE waves, each with DEPTH stages of NL 16-byte loads in flight, the data
// only xor-ed together (no LDS at all).  What the vector memory path of a CU delivers when it is not the LDS-DMA.
template <int NWAVE, int NL, int DEPTH>
__global__ __launch_bounds__(NWAVE * 64) void fetch_l2_k(const unsigned short* __restrict__ A, const unsigned short* __restrict__ W, int M, int N, int K, int BM, int BN, int* sink) {
    const int tid = threadIdx.x;
    const int ntn = N / BN, nwg = gridDim.x;
    const int orig = blockIdx.x, xcd = orig & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    const int m0 = (wg / ntn) * BM, n0 = (wg % ntn) * BN;
    constexpr int NT = NWAVE * 64;
    const int rows = BM + BN, nt = K / 64;
    const unsigned short* src[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int idx = tid + i * NT, row = min(idx / 8, rows - 1), c = idx % 8;
        src[i] = (row < BM ? A + (long)min(m0 + row, M - 1) * K : W + (long)min(n0 + row - BM, N - 1) * K) + c * 8;
    }
    uint4 r[DEPTH][NL];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int i = 0; i < NL; ++i) r[d][i] = *reinterpret_cast<const uint4*>(src[i] + min(d, nt - 1) * 64);
    unsigned acc = 0;
    for (int t = 0; t < nt; t += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
            for (int i = 0; i < NL; ++i) acc ^= r[d][i].x ^ r[d][i].y ^ r[d][i].z ^ r[d][i].w;
            const int tn = min(t + d + DEPTH, nt - 1);
#pragma unroll
            for (int i = 0; i < NL; ++i) r[d][i] = *reinterpret_cast<const uint4*>(src[i] + tn * 64);
        }
    }
    if (acc == 0x7fffffffu) sink[0] = (int)acc;
}

template <int NWAVE, int NL, int DEPTH>
static void run_l2(const char* what, const unsigned short* A, const unsigned short* W, int M, int N, int K, int BM, int BN, int* sink, int NWB, size_t wstride) {
    if ((BM + BN) * 8 > NL * NWAVE * 64) { printf("%s: NL too small\n", what); return; }
    const int grid = (N / BN) * ((M + BM - 1) / BM);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((fetch_l2_k<NWAVE, NL, DEPTH>), dim3(grid), dim3(NWAVE * 64), 0, 0, A, W + (size_t)(i % NWB) * wstride, M, N, K, BM, BN, sink);
    hipEventRecord(a, 0);
    const int iters = 20;
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((fetch_l2_k<NWAVE, NL, DEPTH>), dim3(grid), dim3(NWAVE * 64), 0, 0, A, W + (size_t)(i % NWB) * wstride, M, N, K, BM, BN, sink);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double us = ms * 1e3 / iters, bytes = (double)grid * (BM + BN) * K * 2;
    printf("%-34s M %5d N %4d K %4d  tile %3dx%3d BK 64 to registers only, %d waves x %d loads x depth %d, %3d workgroups: %6.1f us  %5.1f TB/s (%.0f MB)\n", what, M, N, K, BM, BN,
           NWAVE, NL, DEPTH, grid, us, bytes / us * 1e-6, bytes * 1e-6);
}

template <int BK, int NWAVE, int NL>
static void run_reg(const char* what, const unsigned short* A, const unsigned short* W, int M, int N, int K, int BM, int BN, int* sink, int NWB, size_t wstride) {
    const size_t lds = (size_t)2 * (BM + BN) * BK * 2;
    if ((BM + BN) * BK * 2 / 16 > NL * NWAVE * 64) { printf("%s: NL too small\n", what); return; }
    hipFuncSetAttribute((const void*)fetch_reg_k<BK, NWAVE, NL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int grid = (N / BN) * ((M + BM - 1) / BM);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((fetch_reg_k<BK, NWAVE, NL>), dim3(grid), dim3(NWAVE * 64), lds, 0, A, W + (size_t)(i % NWB) * wstride, M, N, K, BM, BN, sink);
    hipEventRecord(a, 0);
    const int iters = 20;
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((fetch_reg_k<BK, NWAVE, NL>), dim3(grid), dim3(NWAVE * 64), lds, 0, A, W + (size_t)(i % NWB) * wstride, M, N, K, BM, BN, sink);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double us = ms * 1e3 / iters, bytes = (double)grid * (BM + BN) * K * 2;
    printf("%-34s M %5d N %4d K %4d  tile %3dx%3d BK %2d via registers, %d waves, %3d workgroups, %5.1f KB LDS: %6.1f us  %5.1f TB/s staged (%.0f MB)\n", what, M, N, K, BM, BN, BK,
           NWAVE, grid, lds / 1024.0, us, bytes / us * 1e-6, bytes * 1e-6);
}

template <int BK, int STAGES, int NWAVE>
static void run(const char* what, const unsigned short* A, const unsigned short* W, int M, int N, int K, int BM, int BN, int* sink, int NWB, size_t wstride) {
    const size_t lds = (size_t)STAGES * (BM + BN) * BK * 2;
    hipFuncSetAttribute((const void*)fetch_k<BK, STAGES, NWAVE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int grid = (N / BN) * ((M + BM - 1) / BM);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((fetch_k<BK, STAGES, NWAVE>), dim3(grid), dim3(NWAVE * 64), lds, 0, A, W + (size_t)(i % NWB) * wstride, M, N, K, BM, BN, sink);
    hipEventRecord(a, 0);
    const int iters = 20;
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((fetch_k<BK, STAGES, NWAVE>), dim3(grid), dim3(NWAVE * 64), lds, 0, A, W + (size_t)(i % NWB) * wstride, M, N, K, BM, BN, sink);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double us = ms * 1e3 / iters, bytes = (double)grid * (BM + BN) * K * 2;
    printf("%-34s M %5d N %4d K %4d  tile %3dx%3d BK %2d x %d stages, %d waves, %3d workgroups, %5.1f KB LDS: %6.1f us  %5.1f TB/s staged (%.0f MB)\n", what, M, N, K, BM, BN, BK, STAGES,
           NWAVE, grid, lds / 1024.0, us, bytes / us * 1e-6, bytes * 1e-6);
}

int main() {
    const int M = 6400, Nmax = 3072, Kmax = 2048, NWB = 24;
    unsigned short *A, *W; int* sink;
    hipMalloc(&A, (size_t)M * Kmax * 2); hipMalloc(&W, (size_t)NWB * Nmax * Kmax * 2); hipMalloc(&sink, 64);
    std::vector<unsigned short> h((size_t)M * Kmax);
    for (auto& x : h) x = 0x3c00 + (rand() & 0x1ff);
    hipMemcpy(A, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    for (int i = 0; i < NWB; ++i) hipMemcpy(W + (size_t)i * Nmax * Kmax, h.data() + i * 4099, (size_t)Nmax * Kmax * 2, hipMemcpyHostToDevice);
    const size_t ws = (size_t)Nmax * Kmax;
    // ff1 (N 2048, K 1024)
    run<32, 3, 8>("ff1 256x128 BK32 (today, 2/CU)", A, W, M, 2048, 1024, 256, 128, sink, NWB, ws);
    run<64, 2, 8>("ff1 256x128 BK64 x2 (1/CU)", A, W, M, 2048, 1024, 256, 128, sink, NWB, ws);
    run<64, 3, 8>("ff1 256x128 BK64 x3 (1/CU)", A, W, M, 2048, 1024, 256, 128, sink, NWB, ws);
    run<32, 3, 8>("ff1 256x256 BK32 x3 (1/CU)", A, W, M, 2048, 1024, 256, 256, sink, NWB, ws);
    run<64, 2, 8>("ff1 256x256 BK64 x2 (1/CU)", A, W, M, 2048, 1024, 256, 256, sink, NWB, ws);
    run<32, 5, 8>("ff1 256x256 BK32 x5 (1/CU)", A, W, M, 2048, 1024, 256, 256, sink, NWB, ws);
    // out (N 1024, K 1024)
    run<32, 3, 4>("out 128x128 BK32 (today, 3/CU)", A, W, M, 1024, 1024, 128, 128, sink, NWB, ws);
    run<64, 2, 4>("out 128x128 BK64 x2 (2/CU)", A, W, M, 1024, 1024, 128, 128, sink, NWB, ws);
    run<32, 3, 8>("out 256x128 BK32 x3", A, W, M, 1024, 1024, 256, 128, sink, NWB, ws);
    run<64, 3, 8>("out 256x128 BK64 x3 (1/CU)", A, W, M, 1024, 1024, 256, 128, sink, NWB, ws);
    // ff2 (N 1024, K 2048)
    run<32, 3, 4>("ff2 128x128 BK32 (today, 3/CU)", A, W, M, 1024, 2048, 128, 128, sink, NWB, ws);
    run<64, 3, 8>("ff2 256x128 BK64 x3 (1/CU)", A, W, M, 1024, 2048, 256, 128, sink, NWB, ws);
    // qkv (N 3072, K 1024)
    run<32, 3, 8>("qkv 320x256 BK32 x3 (today)", A, W, M, 3072, 1024, 320, 256, sink, NWB, ws);
    run<64, 2, 8>("qkv 320x256 BK64 x2", A, W, M, 3072, 1024, 320, 256, sink, NWB, ws);
    { int one = 1; hipMemcpyToSymbol(HIP_SYMBOL(g_same), &one, 4); }
    printf("--- every workgroup fetches the SAME tile (all L2 hits after the first):\n");
    run<64, 2, 8>("ff1 256x256 BK64 x2 same tile", A, W, M, 2048, 1024, 256, 256, sink, NWB, ws);
    run<64, 2, 16>("ff1 256x256 BK64 x2 16 waves same", A, W, M, 2048, 1024, 256, 256, sink, NWB, ws);
    run<32, 3, 8>("ff1 256x128 BK32 x3 same tile", A, W, M, 2048, 1024, 256, 128, sink, NWB, ws);
    { int zero = 0; hipMemcpyToSymbol(HIP_SYMBOL(g_same), &zero, 4); }
    run<64, 2, 16>("ff1 256x256 BK64 x2 16 waves", A, W, M, 2048, 1024, 256, 256, sink, NWB, ws);
    printf("--- L2 -> registers only (no LDS):\n");
    run_l2<16, 4, 2>("ff1 256x256 regs 16w depth 2", A, W, M, 2048, 1024, 256, 256, sink, NWB, ws);
    run_l2<16, 4, 4>("ff1 256x256 regs 16w depth 4", A, W, M, 2048, 1024, 256, 256, sink, NWB, ws);
    run_l2<8, 8, 2>("ff1 256x256 regs 8w depth 2", A, W, M, 2048, 1024, 256, 256, sink, NWB, ws);
    run_l2<8, 8, 4>("ff1 256x256 regs 8w depth 4", A, W, M, 2048, 1024, 256, 256, sink, NWB, ws);
    run_l2<8, 6, 4>("ff1 256x128 regs 8w depth 4 (400)", A, W, M, 2048, 1024, 256, 128, sink, NWB, ws);
    run_l2<8, 6, 4>("out 256x128 regs 8w depth 4", A, W, M, 1024, 1024, 256, 128, sink, NWB, ws);
    // through registers
    run_reg<64, 8, 8>("ff1 256x256 BK64 regs (1/CU)", A, W, M, 2048, 1024, 256, 256, sink, NWB, ws);
    run_reg<64, 4, 16>("ff1 256x256 BK64 regs 4 waves", A, W, M, 2048, 1024, 256, 256, sink, NWB, ws);
    run_reg<64, 8, 6>("ff1 256x128 BK64 regs (400 wg)", A, W, M, 2048, 1024, 256, 128, sink, NWB, ws);
    run_reg<64, 8, 6>("out 256x128 BK64 regs (1/CU)", A, W, M, 1024, 1024, 256, 128, sink, NWB, ws);
    run_reg<64, 4, 12>("out 256x128 BK64 regs 4 waves", A, W, M, 1024, 1024, 256, 128, sink, NWB, ws);
    run_reg<32, 8, 3>("out 256x128 BK32 regs (1/CU)", A, W, M, 1024, 1024, 256, 128, sink, NWB, ws);
    run_reg<64, 8, 6>("ff2 256x128 BK64 regs (1/CU)", A, W, M, 1024, 2048, 256, 128, sink, NWB, ws);
    run_reg<64, 8, 9>("qkv 320x256 BK64 regs", A, W, M, 3072, 1024, 320, 256, sink, NWB, ws);
    run_reg<64, 4, 8>("out 128x128 BK64 regs (400 wg)", A, W, M, 1024, 1024, 128, 128, sink, NWB, ws);
    return 0;
}
