// generated: hand-scheduled radix-2^25.5 multiplication (diagnostic)
__device__ __forceinline__ uint32_t mul19(uint32_t g) { uint32_t t, r; asm("v_lshl_add_u32 %0, %2, 4, %2\n\tv_lshl_add_u32 %1, %2, 1, %0" : "=&v"(t), "=v"(r) : "v"(g)); return r; }
__device__ __forceinline__ void col10(uint64_t &h, uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1, uint32_t a2, uint32_t b2, uint32_t a3, uint32_t b3, uint32_t a4, uint32_t b4, uint32_t a5, uint32_t b5, uint32_t a6, uint32_t b6, uint32_t a7, uint32_t b7, uint32_t a8, uint32_t b8, uint32_t a9, uint32_t b9) {
    uint64_t junk; asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0" : "+v"(h), "=&s"(junk) : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3), "v"(a4), "v"(b4), "v"(a5), "v"(b5), "v"(a6), "v"(b6), "v"(a7), "v"(b7), "v"(a8), "v"(b8), "v"(a9), "v"(b9));
}
__device__ __forceinline__ bpg10::fe10 fe10_mul_asm(const bpg10::fe10 &f, const bpg10::fe10 &g) {
    using namespace bpg10;
    const uint32_t f0 = f.v[0], f1 = f.v[1], f2 = f.v[2], f3 = f.v[3], f4 = f.v[4], f5 = f.v[5], f6 = f.v[6], f7 = f.v[7], f8 = f.v[8], f9 = f.v[9];
    const uint32_t g0 = g.v[0], g1 = g.v[1], g2 = g.v[2], g3 = g.v[3], g4 = g.v[4], g5 = g.v[5], g6 = g.v[6], g7 = g.v[7], g8 = g.v[8], g9 = g.v[9];
    const uint32_t g1_19 = mul19(g1), g2_19 = mul19(g2), g3_19 = mul19(g3), g4_19 = mul19(g4), g5_19 = mul19(g5), g6_19 = mul19(g6), g7_19 = mul19(g7), g8_19 = mul19(g8), g9_19 = mul19(g9);
    const uint32_t f1_2 = 2 * f1, f3_2 = 2 * f3, f5_2 = 2 * f5, f7_2 = 2 * f7, f9_2 = 2 * f9;
    fe10 r; uint64_t h = 0;
    col10(h, f0, g0, f1_2, g9_19, f2, g8_19, f3_2, g7_19, f4, g6_19, f5_2, g5_19, f6, g4_19, f7_2, g3_19, f8, g2_19, f9_2, g1_19);
    r.v[0] = (uint32_t)h & M26; h >>= 26;
    col10(h, f0, g1, f1, g0, f2, g9_19, f3, g8_19, f4, g7_19, f5, g6_19, f6, g5_19, f7, g4_19, f8, g3_19, f9, g2_19);
    r.v[1] = (uint32_t)h & M25; h >>= 25;
    col10(h, f0, g2, f1_2, g1, f2, g0, f3_2, g9_19, f4, g8_19, f5_2, g7_19, f6, g6_19, f7_2, g5_19, f8, g4_19, f9_2, g3_19);
    r.v[2] = (uint32_t)h & M26; h >>= 26;
    col10(h, f0, g3, f1, g2, f2, g1, f3, g0, f4, g9_19, f5, g8_19, f6, g7_19, f7, g6_19, f8, g5_19, f9, g4_19);
    r.v[3] = (uint32_t)h & M25; h >>= 25;
    col10(h, f0, g4, f1_2, g3, f2, g2, f3_2, g1, f4, g0, f5_2, g9_19, f6, g8_19, f7_2, g7_19, f8, g6_19, f9_2, g5_19);
    r.v[4] = (uint32_t)h & M26; h >>= 26;
    col10(h, f0, g5, f1, g4, f2, g3, f3, g2, f4, g1, f5, g0, f6, g9_19, f7, g8_19, f8, g7_19, f9, g6_19);
    r.v[5] = (uint32_t)h & M25; h >>= 25;
    col10(h, f0, g6, f1_2, g5, f2, g4, f3_2, g3, f4, g2, f5_2, g1, f6, g0, f7_2, g9_19, f8, g8_19, f9_2, g7_19);
    r.v[6] = (uint32_t)h & M26; h >>= 26;
    col10(h, f0, g7, f1, g6, f2, g5, f3, g4, f4, g3, f5, g2, f6, g1, f7, g0, f8, g9_19, f9, g8_19);
    r.v[7] = (uint32_t)h & M25; h >>= 25;
    col10(h, f0, g8, f1_2, g7, f2, g6, f3_2, g5, f4, g4, f5_2, g3, f6, g2, f7_2, g1, f8, g0, f9_2, g9_19);
    r.v[8] = (uint32_t)h & M26; h >>= 26;
    col10(h, f0, g9, f1, g8, f2, g7, f3, g6, f4, g5, f5, g4, f6, g3, f7, g2, f8, g1, f9, g0);
    r.v[9] = (uint32_t)h & M25; h >>= 25;
    h = (uint64_t)r.v[0] + h * 19;
    r.v[0] = (uint32_t)h & M26; r.v[1] += (uint32_t)(h >> 26);
    return r;
}
