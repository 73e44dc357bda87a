// Micro-benchmark: issue rate of single VALU opcodes on gfx950 (16 independent chains per lane,
// operands in registers), to price kernels by the instructions they issue.  One line per opcode
// and grid size: lane-ops per clock per CU (128 = one wave64 instruction every 2 cycles on each
// of the 4 SIMD-32s).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int OP>
__global__ __launch_bounds__(256) void op_kernel(const unsigned* in, unsigned* out, int iters, unsigned long long* clk) {
    unsigned a[16];
    for (int i = 0; i < 16; i++) a[i] = in[(threadIdx.x + i * 7) & 1023];
    unsigned b = in[blockIdx.x & 1023];
    const unsigned sb = __builtin_amdgcn_readfirstlane(b);
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a2[8], b2 = {__uint_as_float(b), 1.0f};   // packed-f32 opcodes take register pairs
    for (int i = 0; i < 8; i++) a2[i] = f2{__uint_as_float(a[2 * i]), __uint_as_float(a[2 * i + 1])};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (OP == 0) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 1) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 2) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 3) asm volatile("v_sub_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 4) asm volatile("v_min_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 5) asm volatile("v_max_i32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 6) asm volatile("v_min3_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 7) asm volatile("v_max3_i32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 8) asm volatile("v_med3_i32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 9) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 10) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 11) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a[i]) :);
            if (OP == 12) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(a[i]) :);
            if (OP == 13) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 14) asm volatile("v_or3_b32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 15) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 16) asm volatile("v_bfe_u32 %0, %0, 3, 8" : "+v"(a[i]) :);
            if (OP == 17) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 18) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(a[i]) : "v"(b));
            if (OP == 19) asm volatile("v_alignbyte_b32 %0, %0, %1, 1" : "+v"(a[i]) : "v"(b));
            if (OP == 20) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 21) asm volatile("v_pk_sub_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 22) asm volatile("v_pk_min_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 23) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 24) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 25) asm volatile("v_pk_mad_u16 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 26) asm volatile("v_pk_lshrrev_b16 %0, 1, %0" : "+v"(a[i]) :);
            if (OP == 27) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 28) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 29) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 30) asm volatile("v_dot4_u32_u8 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 31) asm volatile("v_dot2_u32_u16 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 32) asm volatile("v_sad_u8 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 33) asm volatile("v_msad_u8 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 34) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
            if (OP == 35) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
            if (OP == 36) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 37) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) :);
            if (OP == 38) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 39) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 40) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a[i]) :);
            if (OP == 41) asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(a[i]) :);
            if (OP == 42) asm volatile("v_or_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 43) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 44) asm volatile("v_not_b32 %0, %0" : "+v"(a[i]) :);
            if (OP == 45) asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(a[i]) :);
            if (OP == 46) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 47) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 48) asm volatile("v_subrev_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 49) asm volatile("v_max_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 50) asm volatile("v_min_i32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 51) asm volatile("v_max_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 52) asm volatile("v_min_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 53) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 54) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 55) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 56) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 57) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a2[i & 7]) : "v"(b2));
            if (OP == 58) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a2[i & 7]) : "v"(b2));
            if (OP == 59) asm volatile("v_add_u16 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 60) asm volatile("v_sub_u16 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 61) asm volatile("v_max_u16 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 62) asm volatile("v_min_i16 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 63) asm volatile("v_mul_lo_u16 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 64) asm volatile("v_lshlrev_b16 %0, 1, %0" : "+v"(a[i]) :);
            if (OP == 65) asm volatile("v_mul_i32_i24 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 66) asm volatile("v_mad_i32_i24 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 67) asm volatile("v_add_u32_e64 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 68) asm volatile("v_xor_b32_e64 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 69) asm volatile("v_add_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(a[i]) : "v"(b));
            if (OP == 70) asm volatile("v_and_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "+v"(a[i]) : "v"(b));
            if (OP == 71) asm volatile("v_add_u32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b));
            if (OP == 72) asm volatile("v_xor_b32 %0, 0x12345678, %0" : "+v"(a[i]) :);
            if (OP == 73) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "s"(sb));
            if (OP == 74) asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
            if (OP == 75) asm volatile("v_cmp_gt_u32 vcc, %1, %0\n v_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
            if (OP == 76) asm volatile("v_sub_u32 %0, %1, %0\n v_and_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 77) asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 78) asm volatile("v_xad_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 79) asm volatile("v_sad_u16 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 80) asm volatile("v_ffbl_b32 %0, %0" : "+v"(a[i]) :);
            if (OP == 81) asm volatile("v_readfirstlane_b32 s20, %0" : : "v"(a[i]) : "s20");
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    unsigned r = 0;
    for (int i = 0; i < 16; i++) r ^= a[i];
    for (int i = 0; i < 8; i++) r ^= __float_as_uint(a2[i].x) ^ __float_as_uint(a2[i].y);
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int OP>
static void run(const char* name, const unsigned* d_in, unsigned* d_out, unsigned long long* d_clk, int blocks) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((op_kernel<OP>), dim3(blocks), dim3(256), 0, 0, d_in, d_out, 200, d_clk);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((op_kernel<OP>), dim3(blocks), dim3(256), 0, 0, d_in, d_out, iters, d_clk);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long clk[2];
    CHECK(hipMemcpy(clk, d_clk, 16, hipMemcpyDeviceToHost));
    double ops = (double)blocks * 256 * 16 * iters;
    double ghz = (double)clk[0] / ((double)clk[1] / 100e6) / 1e9;
    printf("%-18s blocks=%5d %.3f ms  %.2f Tlane-op/s  clock %.2f GHz  lane-ops/clk/CU %.1f\n", name, blocks, ms,
           ops / ms / 1e9, ghz, ops / (ms * 1e-3) / (ghz * 1e9) / 256.0);
}

int main() {
    unsigned* d_in; unsigned* d_out; unsigned long long* d_clk;
    CHECK(hipMalloc(&d_in, 4096)); CHECK(hipMalloc(&d_out, 4096 * 256 * 4)); CHECK(hipMalloc(&d_clk, 16));
    unsigned h[1024];
    for (int i = 0; i < 1024; i++) h[i] = 2654435761u * (i + 1);
    CHECK(hipMemcpy(d_in, h, 4096, hipMemcpyHostToDevice));
    for (int bpc : {2, 8}) {
        int blocks = 256 * bpc;
        run<0>("v_xor_b32", d_in, d_out, d_clk, blocks);
        run<1>("v_and_b32", d_in, d_out, d_clk, blocks);
        run<2>("v_add_u32", d_in, d_out, d_clk, blocks);
        run<3>("v_sub_u32", d_in, d_out, d_clk, blocks);
        run<4>("v_min_u32", d_in, d_out, d_clk, blocks);
        run<5>("v_max_i32", d_in, d_out, d_clk, blocks);
        run<6>("v_min3_u32", d_in, d_out, d_clk, blocks);
        run<7>("v_max3_i32", d_in, d_out, d_clk, blocks);
        run<8>("v_med3_i32", d_in, d_out, d_clk, blocks);
        run<9>("v_bcnt acc", d_in, d_out, d_clk, blocks);
        run<10>("v_lshl_or_b32", d_in, d_out, d_clk, blocks);
        run<11>("v_lshlrev_b32", d_in, d_out, d_clk, blocks);
        run<12>("v_lshrrev_b32", d_in, d_out, d_clk, blocks);
        run<13>("v_and_or_b32", d_in, d_out, d_clk, blocks);
        run<14>("v_or3_b32", d_in, d_out, d_clk, blocks);
        run<15>("v_add3_u32", d_in, d_out, d_clk, blocks);
        run<16>("v_bfe_u32", d_in, d_out, d_clk, blocks);
        run<17>("v_perm_b32", d_in, d_out, d_clk, blocks);
        run<18>("v_alignbit_b32", d_in, d_out, d_clk, blocks);
        run<19>("v_alignbyte_b32", d_in, d_out, d_clk, blocks);
        run<20>("v_pk_add_u16", d_in, d_out, d_clk, blocks);
        run<21>("v_pk_sub_i16", d_in, d_out, d_clk, blocks);
        run<22>("v_pk_min_i16", d_in, d_out, d_clk, blocks);
        run<23>("v_pk_max_i16", d_in, d_out, d_clk, blocks);
        run<24>("v_pk_mul_lo_u16", d_in, d_out, d_clk, blocks);
        run<25>("v_pk_mad_u16", d_in, d_out, d_clk, blocks);
        run<26>("v_pk_lshrrev_b16", d_in, d_out, d_clk, blocks);
        run<27>("v_mad_u32_u24", d_in, d_out, d_clk, blocks);
        run<28>("v_mul_u32_u24", d_in, d_out, d_clk, blocks);
        run<29>("v_mul_lo_u32", d_in, d_out, d_clk, blocks);
        run<30>("v_dot4_u32_u8", d_in, d_out, d_clk, blocks);
        run<31>("v_dot2_u32_u16", d_in, d_out, d_clk, blocks);
        run<32>("v_sad_u8", d_in, d_out, d_clk, blocks);
        run<33>("v_msad_u8", d_in, d_out, d_clk, blocks);
        run<34>("v_cndmask_b32", d_in, d_out, d_clk, blocks);
        run<35>("v_cmp+cndmask", d_in, d_out, d_clk, blocks);
        run<36>("v_mbcnt_lo", d_in, d_out, d_clk, blocks);
        run<37>("v_mov_b32 dpp", d_in, d_out, d_clk, blocks);
        run<38>("v_add_f32", d_in, d_out, d_clk, blocks);
        run<39>("v_fma_f32", d_in, d_out, d_clk, blocks);
        run<40>("v_cvt_f32_u32", d_in, d_out, d_clk, blocks);
        run<41>("v_cvt_f32_ubyte0", d_in, d_out, d_clk, blocks);
        run<42>("v_or_b32", d_in, d_out, d_clk, blocks);
        run<43>("v_mov_b32", d_in, d_out, d_clk, blocks);
        run<44>("v_not_b32", d_in, d_out, d_clk, blocks);
        run<45>("v_ashrrev_i32", d_in, d_out, d_clk, blocks);
        run<46>("v_lshrrev_b32 v", d_in, d_out, d_clk, blocks);
        run<47>("v_lshlrev_b32 v", d_in, d_out, d_clk, blocks);
        run<48>("v_subrev_u32", d_in, d_out, d_clk, blocks);
        run<49>("v_max_u32", d_in, d_out, d_clk, blocks);
        run<50>("v_min_i32", d_in, d_out, d_clk, blocks);
        run<51>("v_max_f32", d_in, d_out, d_clk, blocks);
        run<52>("v_min_f32", d_in, d_out, d_clk, blocks);
        run<53>("v_mul_f32", d_in, d_out, d_clk, blocks);
        run<54>("v_sub_f32", d_in, d_out, d_clk, blocks);
        run<55>("v_fmac_f32", d_in, d_out, d_clk, blocks);
        run<56>("v_max3_f32", d_in, d_out, d_clk, blocks);
        run<57>("v_pk_fma_f32", d_in, d_out, d_clk, blocks);
        run<58>("v_pk_add_f32", d_in, d_out, d_clk, blocks);
        run<59>("v_add_u16", d_in, d_out, d_clk, blocks);
        run<60>("v_sub_u16", d_in, d_out, d_clk, blocks);
        run<61>("v_max_u16", d_in, d_out, d_clk, blocks);
        run<62>("v_min_i16", d_in, d_out, d_clk, blocks);
        run<63>("v_mul_lo_u16", d_in, d_out, d_clk, blocks);
        run<64>("v_lshlrev_b16", d_in, d_out, d_clk, blocks);
        run<65>("v_mul_i32_i24", d_in, d_out, d_clk, blocks);
        run<66>("v_mad_i32_i24", d_in, d_out, d_clk, blocks);
        run<67>("v_add_u32 e64", d_in, d_out, d_clk, blocks);
        run<68>("v_xor_b32 e64", d_in, d_out, d_clk, blocks);
        run<69>("v_add_u32 sdwa", d_in, d_out, d_clk, blocks);
        run<70>("v_and_b32 sdwa", d_in, d_out, d_clk, blocks);
        run<71>("v_add_u32 dpp", d_in, d_out, d_clk, blocks);
        run<72>("v_xor_b32 lit", d_in, d_out, d_clk, blocks);
        run<73>("v_add_u32 sgpr", d_in, d_out, d_clk, blocks);
        run<74>("v_cmp_gt_u32 vcc", d_in, d_out, d_clk, blocks);
        run<75>("v_cmp+addc", d_in, d_out, d_clk, blocks);
        run<76>("v_sub_u32 x2 + and", d_in, d_out, d_clk, blocks);
        run<77>("v_bfi_b32", d_in, d_out, d_clk, blocks);
        run<78>("v_xad_u32", d_in, d_out, d_clk, blocks);
        run<79>("v_sad_u16", d_in, d_out, d_clk, blocks);
        run<80>("v_ffbl_b32", d_in, d_out, d_clk, blocks);
        run<81>("v_readfirstlane", d_in, d_out, d_clk, blocks);
    }
    return 0;
}
