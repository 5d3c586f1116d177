// k_vit_gemm256, main-loop "schedule 0" (round 2): two barriers per phase, waves 4-7 one segment behind.  Measured against
// schedule 1 (one barrier per phase, the two groups run different programs: what the library ships) in
// profiles/r02_gemm256_microbench_v1.log (1-4 % slower on every shape) and removed from the product in round 3.  It was the
// `if constexpr (VAR == 0)` branch of the kernel body in patchioner_amd/csrc/vit_gemm256.hip; the macros it uses (G256_ISSUE_*,
// G256_READ_*, G256_MMA, G256_BARRIER) are that file's.
  if constexpr (VAR == 0) {
    // ------------------------------------------------------------------------------------------------------------
    // schedule 0: two barriers per phase, waves 4-7 one segment behind (file header)
    // prologue: K-tile 0 whole, A0 / B0 of K-tile 1
    G256_ISSUE_A(0, 0, 0);
    G256_ISSUE_W(0, 0, 0);
    G256_ISSUE_W(0, 1, 0);
    G256_ISSUE_A(0, 1, 0);
    G256_ISSUE_A(1, 0, 1);
    G256_ISSUE_W(1, 0, 1);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    G256_BARRIER();
    G256_STAMP(1);
    if (!PIO_G256_NOSTAGGER && wr == 1) G256_BARRIER();        // waves 4-7 run one segment behind

#define G256_KTILE(t, BUF)                                                                               \
  do {                                                                                                   \
    /* phase 0: (A0, B0) */                                                                              \
    G256_READ_A(BUF, 0);                                                                                 \
    G256_READ_B(fb0, BUF, 0);                                                                            \
    if ((t) + 1 < nk) G256_ISSUE_W((BUF) ^ 1, 1, (t) + 1);                                               \
    G256_BARRIER();                                                                                      \
    G256_MMA(0, 0, fb0);                                                                                 \
    G256_BARRIER();                                                                                      \
    /* phase 1: (A0, B1) */                                                                              \
    G256_READ_B(fb1, BUF, 1);                                                                            \
    if ((t) + 1 < nk) G256_ISSUE_A((BUF) ^ 1, 1, (t) + 1);                                               \
    G256_BARRIER();                                                                                      \
    G256_MMA(0, 1, fb1);                                                                                 \
    G256_BARRIER();                                                                                      \
    /* phase 2: (A1, B1) */                                                                              \
    G256_READ_A(BUF, 1);                                                                                 \
    if ((t) + 2 < nk) G256_ISSUE_A(BUF, 0, (t) + 2);                                                     \
    G256_BARRIER();                                                                                      \
    G256_MMA(1, 1, fb1);                                                                                 \
    G256_BARRIER();                                                                                      \
    /* phase 3: (A1, B0) */                                                                              \
    if ((t) + 2 < nk) {                                                                                  \
      G256_ISSUE_W(BUF, 0, (t) + 2);                                                                     \
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                                   \
    } else {                                                                                             \
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                   \
    }                                                                                                    \
    G256_BARRIER();                                                                                      \
    G256_MMA(1, 0, fb0);                                                                                 \
    G256_BARRIER();                                                                                      \
  } while (0)
#define G256_MAINLOOP(SWAP_)                                                                             \
  do {                                                                                                   \
    constexpr bool SWAP = SWAP_;                                                                         \
    for (int t = 0; t < nk; t += 2) {      /* nk is even (launcher): buffer parity is a literal */       \
      G256_KTILE(t, 0);                                                                                  \
      G256_KTILE(t + 1, 1);                                                                              \
    }                                                                                                    \
  } while (0)

    if (v_block) G256_MAINLOOP(false);
    else G256_MAINLOOP(true);
    if (!PIO_G256_NOSTAGGER && wr == 0) G256_BARRIER();        // re-align: every wave is past its last LDS read
#undef G256_MAINLOOP
#undef G256_KTILE
