/* celt_dec_bands.h — decoder side of the band recursion: quant_all_bands(encode = 0) (celt/bands.c:1589) with
 * quant_band_stereo :1387, quant_band :1248, quant_partition :973, compute_theta :700 (decode branches), alg_unquant
 * (celt/vq.c:621), plus anti_collapse (bands.c:259) and denormalise_bands (bands.c:187).  Same execution model as the
 * encoder: uniform control flow, context by value, range decoder in fenced lane-0 sections, elementwise work on all lanes. */
#ifndef OPUS_AMD_CELT_DEC_BANDS_H
#define OPUS_AMD_CELT_DEC_BANDS_H

/* alg_unquant: cwrsi walks the index with the wave holding one row of U(n,k) per step (celt_dec_energy.h), everything after it is elementwise / in registers */
template <int NR> WV_DEV unsigned alg_unquant_regs(WV_LDS DecLds *L, WV_LDS i32 *X, int N, int K, int spread, int B, i32 gain)
{
   const int lane = wv_lane();
   WV_LDS i32 *iy = L->BC.q.iy;
   const u32 ft = pvq_u(N, K) + pvq_u(N, K + 1);
   wv_sync();
   LANE0 { EC_BEGIN; L->sh.r[0] = (i32)k_ec_dec_uint(EC_PASS, ft); EC_END; }
   const i32 Ryy = cwrsi_wave(N, K, (u32)wv_uni(L->sh.r[0]), iy);
   i32 v[NR], q[NR];
   for (int t = 0; t < NR; t++) q[t] = lane + 64 * t < N ? iy[lane + 64 * t] : 0;
   int k = celt_ilog2(Ryy) >> 1;
   i32 t_ = vshr32(Ryy, 2 * (k - 7) - 15);
   i32 g = mult32_32_q31(fx_rsqrt_norm32(t_), gain);
   for (int t = 0; t < NR; t++) v[t] = vshr32(mult16_32_q15(q[t], g), k + 15 - NORM_SHIFT);
   exp_rotation_regs(v, X, N, -1, B, K, spread);                 /* X: the output slot, free as scratch until the store below */
   unsigned cm = 1;
   if (B > 1) {
      int N0 = (u32)N / (u32)B;
      u32 m = 0;
      for (int t = 0; t < NR; t++) if (q[t] != 0) m |= 1u << ((u32)(lane + 64 * t) / (u32)N0);
      cm = wv_or(m);
   }
   wv_sync();
   for (int t = 0; t < NR; t++) if (lane + 64 * t < N) X[lane + 64 * t] = v[t];
   wv_sync();
   return cm;
}
WV_DEVN unsigned alg_unquant_wave(WV_LDS DecLds *L, WV_LDS i32 *X, int N, int K, int spread, int B, i32 gain)
{
   N = wv_uni(N); K = wv_uni(K); spread = wv_uni(spread); B = wv_uni(B); gain = wv_uni(gain);
   if (N <= 64) return alg_unquant_regs<1>(L, X, N, K, spread, B, gain);
   return alg_unquant_regs<3>(L, X, N, K, spread, B, gain);
}

/* compute_theta, decode side.  Returns {inv, imid, iside, delta, itheta, qalloc, b, fill}. */
WV_DEVN i32x8 dec_compute_theta_wave(WV_LDS DecLds *L, BandCfg cfg, i32 remaining_bits, int N, int b, int B, int B0, int LM, int stereo, int fill)
{
   cfg = cfg_uni(cfg); remaining_bits = wv_uni(remaining_bits); N = wv_uni(N); b = wv_uni(b); B = wv_uni(B); B0 = wv_uni(B0); LM = wv_uni(LM);
   stereo = wv_uni(stereo); fill = wv_uni(fill);
   int qn, itheta = 0, delta, imid, iside, qalloc, pulse_cap, offset, inv = 0;
   const int i = cfg.i, intensity = cfg.intensity;
   pulse_cap = ct_logN[i] + LM * (1 << BITRES);
   offset = (pulse_cap >> 1) - (stereo && N == 2 ? 16 : 4);
   qn = compute_qn(N, b, offset, pulse_cap, stereo);
   if (stereo && i >= intensity) qn = 1;
   wv_sync();
   i32 tell = ec_tell_frac_lds(&L->ec);
   wv_sync();
   if (qn != 1) {
      LANE0 {
         EC_BEGIN;
         int it;
         if (stereo && N > 2) {
            int p0 = 3, x, x0 = qn / 2, ft = p0 * (x0 + 1) + x0;
            int fs = k_ec_decode(EC_PASS, ft);
            if (fs < (x0 + 1) * p0) x = fs / p0;
            else x = x0 + 1 + (fs - (x0 + 1) * p0);
            k_ec_dec_update(EC_PASS, x <= x0 ? p0 * x : (x - 1 - x0) + (x0 + 1) * p0, x <= x0 ? p0 * (x + 1) : (x - x0) + (x0 + 1) * p0, ft);
            it = x;
         } else if (B0 > 1 || stereo) {
            it = k_ec_dec_uint(EC_PASS, qn + 1);
         } else {
            int fs, fl, ft = ((qn >> 1) + 1) * ((qn >> 1) + 1);
            int fm = k_ec_decode(EC_PASS, ft);
            if (fm < ((qn >> 1) * ((qn >> 1) + 1) >> 1)) {
               it = (fx_isqrt32(8 * (u32)fm + 1) - 1) >> 1;
               fs = it + 1;
               fl = it * (it + 1) >> 1;
            } else {
               it = (2 * (qn + 1) - fx_isqrt32(8 * (u32)(ft - fm - 1) + 1)) >> 1;
               fs = qn + 1 - it;
               fl = ft - ((qn + 1 - it) * (qn + 2 - it) >> 1);
            }
            k_ec_dec_update(EC_PASS, fl, fl + fs, ft);
         }
         L->sh.r[0] = it;
         EC_END;
      }
      itheta = wv_uni(L->sh.r[0]);
      itheta = (u32)((i32)itheta * 16384) / (u32)qn;
   } else if (stereo) {
      if (b > 2 << BITRES && remaining_bits > 2 << BITRES) {
         LANE0 { EC_BEGIN; L->sh.r[0] = k_ec_dec_bit_logp(EC_PASS, 2); EC_END; }
         inv = wv_uni(L->sh.r[0]);
      } else inv = 0;
      if (cfg.disable_inv) inv = 0;
      itheta = 0;
   }
   wv_sync();
   qalloc = ec_tell_frac_lds(&L->ec) - tell;
   b -= qalloc;
   if (itheta == 0) { imid = 32767; iside = 0; fill &= (1 << B) - 1; delta = -16384; }
   else if (itheta == 16384) { imid = 0; iside = 32767; fill &= ((1 << B) - 1) << B; delta = 16384; }
   else {
      imid = bitexact_cos((i16)itheta);
      iside = bitexact_cos((i16)(16384 - itheta));
      delta = frac_mul16((N - 1) << 7, bitexact_log2tan(iside, imid));
   }
   i32x8 r = {inv, imid, iside, delta, itheta, qalloc, b, fill};
   return r;
}

WV_DEV unsigned dec_quant_band_n1_wave(WV_LDS DecLds *L, i32 &remaining_bits, WV_LDS i32 *X, WV_LDS i32 *Y, WV_LDS i32 *lowband_out)
{
   WV_LDS i32 *x = X;
   int stereo = Y != 0;
   wv_sync();
   for (int c = 0; c < 1 + stereo; c++) {
      int sign = 0;
      if (remaining_bits >= 1 << BITRES) {
         LANE0 { EC_BEGIN; L->sh.r[0] = k_ec_dec_bits(EC_PASS, 1); EC_END; }
         sign = wv_uni(L->sh.r[0]);
         remaining_bits -= 1 << BITRES;
      }
      wv_sync(); LANE0 x[0] = sign ? -(1 << NORM_SHIFT) : (1 << NORM_SHIFT); wv_sync();
      x = Y;
   }
   wv_sync();
   if (lowband_out) { LANE0 lowband_out[0] = X[0] >> 4; }
   wv_sync();
   return 1;
}

/* (as in the encoder: the body is inlined at its single depth-0 call site, deeper levels are out-of-line instantiations) */
template <int DEPTH> WV_DEVN i32x4 dec_quant_partition_wave(WV_LDS DecLds *L, BandCfg cfg, i32 remaining_bits, u32 seed, WV_LDS i32 *X, int N, int b, int B, WV_LDS i32 *lowband,
      int LM, i32 gain, int fill);
template <int DEPTH>
WV_DEV i32x4 dec_quant_partition_body(WV_LDS DecLds *L, BandCfg cfg, i32 remaining_bits, u32 seed, WV_LDS i32 *X, int N, int b, int B, WV_LDS i32 *lowband,
      int LM, i32 gain, int fill)
{
   cfg = cfg_uni(cfg); remaining_bits = wv_uni(remaining_bits); seed = (u32)wv_uni((i32)seed); N = wv_uni(N); b = wv_uni(b); B = wv_uni(B);
   LM = wv_uni(LM); gain = wv_uni(gain); fill = wv_uni(fill);
   int B0 = B;
   const int i = cfg.i, spread = cfg.spread;
   unsigned cm = 0;
   const i32 row = cache_row_load(i, LM);
   bool split = LM != -1 && b > wv_bcast(row, wv_bcast(row, 0)) + 12 && N > 2;
   if constexpr (DEPTH < 4) {
      if (split) {
         int mbits, sbits, delta, itheta, qalloc;
         WV_LDS i32 *next_lowband2 = 0, *Y;
         i32 rebalance, mid, side;
         N >>= 1;
         Y = X + N;
         LM -= 1;
         if (B == 1) fill = (fill & 1) | (fill << 1);
         B = (B + 1) >> 1;
         const i32x8 th = dec_compute_theta_wave(L, cfg, remaining_bits, N, b, B, B0, LM, 0, fill);
         delta = wv_uni(th[3]); itheta = wv_uni(th[4]); qalloc = wv_uni(th[5]); b = wv_uni(th[6]); fill = wv_uni(th[7]);
         mid = shl32((i32)wv_uni(th[1]), 16);
         side = shl32((i32)wv_uni(th[2]), 16);
         if (B0 > 1 && (itheta & 0x3fff)) {
            if (itheta > 8192) delta -= delta >> (4 - LM);
            else delta = imin(0, delta + (N << BITRES >> (5 - LM)));
         }
         mbits = imax(0, imin(b, (b - delta) / 2));
         sbits = b - mbits;
         remaining_bits -= qalloc;
         if (lowband) next_lowband2 = lowband + N;
         rebalance = remaining_bits;
         i32x4 r;
         if (mbits >= sbits) {
            r = dec_quant_partition_wave<DEPTH + 1>(L, cfg, remaining_bits, seed, X, N, mbits, B, lowband, LM, mult32_32_q31(gain, mid), fill);
            cm = (unsigned)wv_uni(r[0]); remaining_bits = wv_uni(r[1]); seed = (u32)wv_uni(r[2]);
            rebalance = mbits - (rebalance - remaining_bits);
            if (rebalance > 3 << BITRES && itheta != 0) sbits += rebalance - (3 << BITRES);
            r = dec_quant_partition_wave<DEPTH + 1>(L, cfg, remaining_bits, seed, Y, N, sbits, B, next_lowband2, LM, mult32_32_q31(gain, side), fill >> B);
            cm |= (unsigned)wv_uni(r[0]) << (B0 >> 1); remaining_bits = wv_uni(r[1]); seed = (u32)wv_uni(r[2]);
         } else {
            r = dec_quant_partition_wave<DEPTH + 1>(L, cfg, remaining_bits, seed, Y, N, sbits, B, next_lowband2, LM, mult32_32_q31(gain, side), fill >> B);
            cm = (unsigned)wv_uni(r[0]) << (B0 >> 1); remaining_bits = wv_uni(r[1]); seed = (u32)wv_uni(r[2]);
            rebalance = sbits - (rebalance - remaining_bits);
            if (rebalance > 3 << BITRES && itheta != 16384) mbits += rebalance - (3 << BITRES);
            r = dec_quant_partition_wave<DEPTH + 1>(L, cfg, remaining_bits, seed, X, N, mbits, B, lowband, LM, mult32_32_q31(gain, mid), fill);
            cm |= (unsigned)wv_uni(r[0]); remaining_bits = wv_uni(r[1]); seed = (u32)wv_uni(r[2]);
         }
         return ret3(cm, remaining_bits, seed);
      }
   }
   {
      int q = row_bits2pulses(row, b);
      int curr_bits = row_pulses2bits(row, q);
      remaining_bits -= curr_bits;
      while (remaining_bits < 0 && q > 0) {
         remaining_bits += curr_bits;
         q--;
         curr_bits = row_pulses2bits(row, q);
         remaining_bits -= curr_bits;
      }
      if (q != 0) {
         int K = k_get_pulses(q);
         cm = alg_unquant_wave(L, X, N, K, spread, B, gain);
      } else {
         unsigned cm_mask = (unsigned)(1UL << B) - 1;
         fill &= cm_mask;
         if (!fill) { wv_sync(); FOR_LANES(j, N) X[j] = 0; wv_sync(); }
         else {
            wv_sync();
            if (lowband == 0) {
               LANE0 { u32 s = seed; for (int j = 0; j < N; j++) { s = lcg_rand(s); X[j] = shl32((i32)((i32)s >> 20), NORM_SHIFT - 14); } }
               cm = cm_mask;
            } else {
               LANE0 {
                  u32 s = seed;
                  for (int j = 0; j < N; j++) {
                     s = lcg_rand(s);
                     i16 tmp = QC16(1.0f / 256, NORM_SHIFT - 4);
                     tmp = (s) & 0x8000 ? tmp : -tmp;
                     X[j] = lowband[j] + tmp;
                  }
               }
               cm = fill;
            }
            for (int j = 0; j < N; j++) seed = lcg_rand(seed);
            wv_sync();
            renormalise_vector_wave(X, N, gain);
         }
      }
   }
   return ret3(cm, remaining_bits, seed);
}

template <int DEPTH> WV_DEVN i32x4 dec_quant_partition_wave(WV_LDS DecLds *L, BandCfg cfg, i32 remaining_bits, u32 seed, WV_LDS i32 *X, int N, int b, int B, WV_LDS i32 *lowband,
      int LM, i32 gain, int fill)
{
   return dec_quant_partition_body<DEPTH>(L, cfg, remaining_bits, seed, X, N, b, B, lowband, LM, gain, fill);
}

WV_DEVN i32x4 dec_quant_band_wave(WV_LDS DecLds *L, BandCfg cfg, i32 remaining_bits, u32 seed, WV_LDS i32 *X, int N, int b, int B, WV_LDS i32 *lowband, int LM,
      WV_LDS i32 *lowband_out, i32 gain, WV_LDS i32 *lowband_scratch, int fill)
{
   cfg = cfg_uni(cfg); remaining_bits = wv_uni(remaining_bits); seed = (u32)wv_uni((i32)seed); N = wv_uni(N); b = wv_uni(b); B = wv_uni(B);
   LM = wv_uni(LM); gain = wv_uni(gain); fill = wv_uni(fill);
   const u8 bit_interleave_table[16] = {0, 1, 1, 1, 2, 3, 3, 3, 2, 3, 3, 3, 2, 3, 3, 3};
   const u8 bit_deinterleave_table[16] = {0x00, 0x03, 0x0C, 0x0F, 0x30, 0x33, 0x3C, 0x3F, 0xC0, 0xC3, 0xCC, 0xCF, 0xF0, 0xF3, 0xFC, 0xFF};
   int N0 = N, N_B = N, N_B0, B0 = B, time_divide = 0, recombine = 0, longBlocks, k;
   unsigned cm = 0;
   int tf_change = cfg.tf_change;
   longBlocks = B0 == 1;
   N_B = (u32)N_B / (u32)B;
   if (N == 1) { cm = dec_quant_band_n1_wave(L, remaining_bits, X, 0, lowband_out); return ret3(cm, remaining_bits, seed); }
   if (tf_change > 0) recombine = tf_change;
   if (lowband_scratch && lowband && (recombine || ((N_B & 1) == 0 && tf_change < 0) || B0 > 1)) {
      wv_sync();
      FOR_LANES(j, N) lowband_scratch[j] = lowband[j];
      wv_sync();
      lowband = lowband_scratch;
   }
   for (k = 0; k < recombine; k++) {
      if (lowband) haar1_wave(lowband, N >> k, 1 << k);
      fill = bit_interleave_table[fill & 0xF] | bit_interleave_table[fill >> 4] << 2;
   }
   B >>= recombine;
   N_B <<= recombine;
   while ((N_B & 1) == 0 && tf_change < 0) {
      if (lowband) haar1_wave(lowband, N_B, B);
      fill |= fill << B;
      B <<= 1;
      N_B >>= 1;
      time_divide++;
      tf_change++;
   }
   B0 = B;
   N_B0 = N_B;
   if (B0 > 1 && lowband) deinterleave_hadamard_wave(lowband, N_B >> recombine, B0 << recombine, longBlocks);
   {
      const i32x4 r = dec_quant_partition_body<0>(L, cfg, remaining_bits, seed, X, N, b, B, lowband, LM, gain, fill);
      cm = (unsigned)wv_uni(r[0]); remaining_bits = wv_uni(r[1]); seed = (u32)wv_uni(r[2]);
   }
   {
      if (B0 > 1) interleave_hadamard_wave(X, N_B >> recombine, B0 << recombine, longBlocks);
      N_B = N_B0;
      B = B0;
      for (k = 0; k < time_divide; k++) {
         B >>= 1;
         N_B <<= 1;
         cm |= cm >> B;
         haar1_wave(X, N_B, B);
      }
      for (k = 0; k < recombine; k++) {
         cm = bit_deinterleave_table[cm];
         haar1_wave(X, N0 >> k, 1 << k);
      }
      B <<= recombine;
      if (lowband_out) {
         i16 n = (i16)fx_sqrt(shl32((i32)N0, 22));
         wv_sync();
         FOR_LANES(j, N0) lowband_out[j] = mult16_32_q15(n, X[j]);
         wv_sync();
      }
      cm &= (1 << B) - 1;
   }
   return ret3(cm, remaining_bits, seed);
}

WV_DEV i32x4 dec_quant_band_stereo_wave(WV_LDS DecLds *L, BandCfg cfg, i32 remaining_bits, u32 seed, WV_LDS i32 *X, WV_LDS i32 *Y, int N, int b, int B,
      WV_LDS i32 *lowband, int LM, WV_LDS i32 *lowband_out, WV_LDS i32 *lowband_scratch, int fill)
{
   cfg = cfg_uni(cfg); remaining_bits = wv_uni(remaining_bits); seed = (u32)wv_uni((i32)seed); N = wv_uni(N); b = wv_uni(b); B = wv_uni(B);
   LM = wv_uni(LM); fill = wv_uni(fill);
   int inv = 0, mbits, sbits, delta, itheta, qalloc, orig_fill;
   i32 mid = 0, side = 0;
   unsigned cm = 0;
   i32x4 r;
   if (N == 1) { cm = dec_quant_band_n1_wave(L, remaining_bits, X, Y, lowband_out); return ret3(cm, remaining_bits, seed); }
   orig_fill = fill;
   {
      const i32x8 th = dec_compute_theta_wave(L, cfg, remaining_bits, N, b, B, B, LM, 1, fill);
      inv = wv_uni(th[0]); delta = wv_uni(th[3]); itheta = wv_uni(th[4]); qalloc = wv_uni(th[5]); b = wv_uni(th[6]); fill = wv_uni(th[7]);
      mid = shl32((i32)wv_uni(th[1]), 16);
      side = shl32((i32)wv_uni(th[2]), 16);
   }
   if (N == 2) {
      int c, sign = 0;
      WV_LDS i32 *x2, *y2;
      mbits = b;
      sbits = 0;
      if (itheta != 0 && itheta != 16384) sbits = 1 << BITRES;
      mbits -= sbits;
      c = itheta > 8192;
      remaining_bits -= qalloc + sbits;
      x2 = c ? Y : X;
      y2 = c ? X : Y;
      wv_sync();
      if (sbits) {
         LANE0 { EC_BEGIN; L->sh.r[0] = k_ec_dec_bits(EC_PASS, 1); EC_END; }
         sign = wv_uni(L->sh.r[0]);
      }
      sign = 1 - 2 * sign;
      r = dec_quant_band_wave(L, cfg, remaining_bits, seed, x2, N, mbits, B, lowband, LM, lowband_out, Q31ONE, lowband_scratch, orig_fill);
      cm = (unsigned)wv_uni(r[0]); remaining_bits = wv_uni(r[1]); seed = (u32)wv_uni(r[2]);
      wv_sync();
      LANE0 { y2[0] = -sign * x2[1]; y2[1] = sign * x2[0]; }
      wv_sync();
      LANE0 {
         i32 tmp;
         X[0] = mult32_32_q31(mid, X[0]);
         X[1] = mult32_32_q31(mid, X[1]);
         Y[0] = mult32_32_q31(side, Y[0]);
         Y[1] = mult32_32_q31(side, Y[1]);
         tmp = X[0]; X[0] = sub32(tmp, Y[0]); Y[0] = add32(tmp, Y[0]);
         tmp = X[1]; X[1] = sub32(tmp, Y[1]); Y[1] = add32(tmp, Y[1]);
      }
      wv_sync();
   } else {
      i32 rebalance;
      mbits = imax(0, imin(b, (b - delta) / 2));
      sbits = b - mbits;
      remaining_bits -= qalloc;
      rebalance = remaining_bits;
      if (mbits >= sbits) {
         r = dec_quant_band_wave(L, cfg, remaining_bits, seed, X, N, mbits, B, lowband, LM, lowband_out, Q31ONE, lowband_scratch, fill);
         cm = (unsigned)wv_uni(r[0]); remaining_bits = wv_uni(r[1]); seed = (u32)wv_uni(r[2]);
         rebalance = mbits - (rebalance - remaining_bits);
         if (rebalance > 3 << BITRES && itheta != 0) sbits += rebalance - (3 << BITRES);
         r = dec_quant_band_wave(L, cfg, remaining_bits, seed, Y, N, sbits, B, 0, LM, 0, side, 0, fill >> B);
         cm |= (unsigned)wv_uni(r[0]); remaining_bits = wv_uni(r[1]); seed = (u32)wv_uni(r[2]);
      } else {
         r = dec_quant_band_wave(L, cfg, remaining_bits, seed, Y, N, sbits, B, 0, LM, 0, side, 0, fill >> B);
         cm = (unsigned)wv_uni(r[0]); remaining_bits = wv_uni(r[1]); seed = (u32)wv_uni(r[2]);
         rebalance = sbits - (rebalance - remaining_bits);
         if (rebalance > 3 << BITRES && itheta != 16384) mbits += rebalance - (3 << BITRES);
         r = dec_quant_band_wave(L, cfg, remaining_bits, seed, X, N, mbits, B, lowband, LM, lowband_out, Q31ONE, lowband_scratch, fill);
         cm |= (unsigned)wv_uni(r[0]); remaining_bits = wv_uni(r[1]); seed = (u32)wv_uni(r[2]);
      }
   }
   if (N != 2) stereo_merge_wave(X, Y, mid, N);
   if (inv) { wv_sync(); FOR_LANES(j, N) Y[j] = neg32(Y[j]); wv_sync(); }
   return ret3(cm, remaining_bits, seed);
}

/* quant_all_bands(encode = 0) */
WV_DEVN void dec_quant_all_bands_wave(WV_LDS DecLds *L, int shortBlocks, int spread, int dual_stereo, int intensity, i32 total_bits, i32 balance,
      int codedBands, int disable_inv)
{
   shortBlocks = wv_uni(shortBlocks); spread = wv_uni(spread); dual_stereo = wv_uni(dual_stereo); intensity = wv_uni(intensity); total_bits = wv_uni(total_bits);
   balance = wv_uni(balance); codedBands = wv_uni(codedBands); disable_inv = wv_uni(disable_inv);
   const int start = wv_uni(L->sh.start), end = wv_uni(L->sh.end), LM = wv_uni(L->sh.LM), C = wv_uni(L->sh.C), Nfull = wv_uni(L->sh.N);
   i32 *Xg = L->Xg, *Yg = C == 2 ? L->Xg + Nfull : 0;                                  /* the spectrum in the wave's HBM scratch; a band is decoded in LDS (xb / yb) and copied out */
   /* the folding memory (bands.c:1600 `norm`, `norm2`) sits behind the spectrum in the wave's HBM scratch: a band's source is staged into `lbs` before the band is decoded (the staging copy
    * doubles as quant_band's lowband scratch), its contribution leaves through `lbo` when it is done -- both lane-parallel and coalesced */
   i32 *norm = L->Xg + 2 * OA_MAX_FRAME, *norm2 = norm + OA_NORM_LEN;
   WV_LDS i32 *const lbs = L->BC.q.lbs, *const lbo_buf = L->BC.q.lbo;
   WV_LDS u8 *collapse_masks = L->collapse_masks;
   const WV_LDS i32 *pulses = L->pulses, *tf_res = L->tf_res;
   i32 remaining_bits;
   int M = 1 << LM, B = shortBlocks ? M : 1, lowband_offset = 0, update_lowband = 1;
   int norm_offset = M * ct_eBands[start];
   /* (the reference borrows the last band of X as lowband scratch, bands.c:1642-1653, and stops using it when it decodes that band: a buffer of its own here) */
   WV_LDS i32 *lowband_scratch = lbs;
   BandCfg cfg;
   u32 seed = (u32)wv_uni((i32)L->st.rng);
   i32x4 r;
   cfg.intensity = intensity; cfg.spread = spread; cfg.disable_inv = disable_inv; cfg.resynth = 1;
   cfg.theta_round = 0; cfg.avoid_split_noise = B > 1; cfg.i = 0; cfg.tf_change = 0;
   for (int i = start; i < end; i++) {
      i32 tell, curr_balance;
      int b, N, effective_lowband = -1, tf_change = 0, last;
      WV_LDS i32 *X, *Y;
      unsigned x_cm, y_cm;
      cfg.i = i;
      last = (i == end - 1);
      X = L->BC.q.xb;
      Y = Yg != 0 ? L->BC.q.yb : 0;
      N = M * ct_eBands[i + 1] - M * ct_eBands[i];
      wv_sync();
      tell = wv_uni(ec_tell_frac_lds(&L->ec));
      if (i != start) balance -= tell;
      remaining_bits = total_bits - tell - 1;
      if (i <= codedBands - 1) {
         curr_balance = balance / imin(3, codedBands - i);
         b = imax(0, imin(16383, imin(remaining_bits + 1, wv_uni(pulses[i]) + curr_balance)));
      } else b = 0;
      if ((M * ct_eBands[i] - N >= M * ct_eBands[start] || i == start + 1) && (update_lowband || lowband_offset == 0))
         lowband_offset = i;
      if (i == start + 1) {            /* special_hybrid_folding (bands.c:1575, RFC 8251 section 9): the second band never has to fold from the LCG; copies nothing when start == 0 */
         const int n1 = M * (ct_eBands[start + 1] - ct_eBands[start]), n2 = M * (ct_eBands[start + 2] - ct_eBands[start + 1]);
         wv_sync();
         FOR_LANES(j, n2 - n1) { norm[n1 + j] = norm[2 * n1 - n2 + j]; if (dual_stereo) norm2[n1 + j] = norm2[2 * n1 - n2 + j]; }
         wv_sync();
      }
      tf_change = wv_uni(tf_res[i]);
      cfg.tf_change = tf_change;
      if (last) lowband_scratch = 0;
      if (lowband_offset != 0 && (spread != 3 || B > 1 || tf_change < 0)) {
         int fold_start, fold_end, fold_i;
         effective_lowband = imax(0, M * ct_eBands[lowband_offset] - norm_offset - N);
         fold_start = lowband_offset;
         while (M * ct_eBands[--fold_start] > effective_lowband + norm_offset);
         fold_end = lowband_offset - 1;
         while (++fold_end < i && M * ct_eBands[fold_end] < effective_lowband + norm_offset + N);
         x_cm = y_cm = 0;
         fold_i = fold_start;
         do {
            x_cm |= collapse_masks[fold_i * C + 0];
            y_cm |= collapse_masks[fold_i * C + C - 1];
         } while (++fold_i < fold_end);
         x_cm = (unsigned)wv_uni((i32)x_cm); y_cm = (unsigned)wv_uni((i32)y_cm);
      } else x_cm = y_cm = (1 << B) - 1;
      if (dual_stereo && i == intensity) {
         dual_stereo = 0;
         wv_sync(); FOR_LANES(j, M * ct_eBands[i] - norm_offset) norm[j] = half32(norm[j] + norm2[j]); wv_sync();
      }
      WV_LDS i32 *const lb = effective_lowband != -1 ? lbs : 0;
      WV_LDS i32 *const lbo = last ? 0 : lbo_buf;
      const int out_at = M * ct_eBands[i] - norm_offset;
      wv_sync();
      if (lb) { FOR_LANES(j, N) lbs[j] = norm[effective_lowband + j]; }
      wv_sync();
      if (dual_stereo) {
         r = dec_quant_band_wave(L, cfg, remaining_bits, seed, X, N, b / 2, B, lb, LM, lbo, Q31ONE, lowband_scratch, x_cm);
         x_cm = (unsigned)wv_uni(r[0]); remaining_bits = wv_uni(r[1]); seed = (u32)wv_uni(r[2]);
         wv_sync();
         if (lbo) { FOR_LANES(j, N) norm[out_at + j] = lbo_buf[j]; }
         if (lb) { FOR_LANES(j, N) lbs[j] = norm2[effective_lowband + j]; }
         wv_sync();
         r = dec_quant_band_wave(L, cfg, remaining_bits, seed, Y, N, b / 2, B, lb, LM, lbo, Q31ONE, lowband_scratch, y_cm);
         y_cm = (unsigned)wv_uni(r[0]); remaining_bits = wv_uni(r[1]); seed = (u32)wv_uni(r[2]);
         wv_sync();
         if (lbo) { FOR_LANES(j, N) norm2[out_at + j] = lbo_buf[j]; }
      } else {
         if (Y != 0) {
            cfg.theta_round = 0;
            r = dec_quant_band_stereo_wave(L, cfg, remaining_bits, seed, X, Y, N, b, B, lb, LM, lbo, lowband_scratch, x_cm | y_cm);
         } else {
            r = dec_quant_band_wave(L, cfg, remaining_bits, seed, X, N, b, B, lb, LM, lbo, Q31ONE, lowband_scratch, x_cm | y_cm);
         }
         x_cm = (unsigned)wv_uni(r[0]); remaining_bits = wv_uni(r[1]); seed = (u32)wv_uni(r[2]);
         y_cm = x_cm;
         wv_sync();
         if (lbo) { FOR_LANES(j, N) norm[out_at + j] = lbo_buf[j]; }
      }
      wv_sync();
      FOR_LANES(j, N) { Xg[M * ct_eBands[i] + j] = X[j]; if (Y != 0) Yg[M * ct_eBands[i] + j] = Y[j]; }       /* the finished band -> the spectrum */
      LANE0 { collapse_masks[i * C + 0] = (u8)x_cm; collapse_masks[i * C + C - 1] = (u8)y_cm; }
      balance += wv_uni(pulses[i]) + tell;
      update_lowband = b > (N << BITRES);
      cfg.avoid_split_noise = 0;
   }
   wv_sync();
   LANE0 L->st.rng = seed;
   wv_sync();
}

/* anti_collapse (bands.c:259), decoder call.  The LCG advances once per injected sample in (band, channel, k, j) order, so the
 * noise is generated by lane 0; the renormalisation that follows is elementwise. */
WV_DEVN void anti_collapse_wave(WV_LDS DecLds *L, int LM, int C, int size, int start, int end)
{
   i32 *X_ = L->Xg;
   const WV_LDS i32 *logE = L->oldBandE, *prev1logE = L->oldLogE, *prev2logE = L->oldLogE2;
   u32 seed = (u32)wv_uni((i32)L->st.rng);
   for (int i = start; i < end; i++) {
      int N0 = ct_eBands[i + 1] - ct_eBands[i];
      int depth = (int)((u32)(1 + wv_uni(L->pulses[i])) / (u32)N0) >> LM;
      i32 thresh32 = fx_exp2(-shl16(depth, 10 - BITRES)) >> 1;
      i16 thresh = (i16)mult16_32_q15(QC16(0.5f, 15), imin(32767, thresh32));
      i32 t = N0 << LM;
      int shift = celt_ilog2(t) >> 1;
      t = shl32(t, (7 - shift) << 1);
      i16 sqrt_1 = fx_rsqrt_norm(t);
      for (int c = 0; c < C; c++) {
         i32 prev1 = wv_uni(prev1logE[c * NBE + i]), prev2 = wv_uni(prev2logE[c * NBE + i]);
         int renormalize = 0;
         if (C == 1) { prev1 = imax(prev1, wv_uni(prev1logE[NBE + i])); prev2 = imax(prev2, wv_uni(prev2logE[NBE + i])); }
         i32 Ediff = wv_uni(logE[c * NBE + i]) - imin(prev1, prev2);
         Ediff = imax(0, Ediff);
         i32 r;
         if (Ediff < GC(16.f)) { i32 r32 = fx_exp2_db(-Ediff) >> 1; r = 2 * imin(16383, r32); }
         else r = 0;
         if (LM == 3) r = mult16_16_q14(23170, imin(23169, r));
         r = (i16)(imin(thresh, r)) >> 1;
         r = vshr32(mult16_16_q15(sqrt_1, r), shift + 14 - NORM_SHIFT);
         i32 *Xs = X_ + c * size + (ct_eBands[i] << LM);
         WV_LDS i32 *X = L->BC.q.xb;                                       /* a band that needs filling goes through the staging buffer (PVQ phase memory: free by now) */
         const unsigned mask = (unsigned)wv_uni((i32)L->collapse_masks[i * C + c]);
         if ((mask & ((1u << (1 << LM)) - 1)) != ((1u << (1 << LM)) - 1)) { wv_sync(); FOR_LANES(j, N0 << LM) X[j] = Xs[j]; }
         for (int k = 0; k < 1 << LM; k++) {
            if (!(mask & 1 << k)) {
               wv_sync();
               LANE0 { u32 s = seed; for (int j = 0; j < N0; j++) { s = lcg_rand(s); X[(j << LM) + k] = (s & 0x8000 ? r : -r); } }
               for (int j = 0; j < N0; j++) seed = lcg_rand(seed);
               renormalize = 1;
            }
         }
         if (renormalize) { wv_sync(); renormalise_vector_wave(X, N0 << LM, Q31ONE); wv_sync(); FOR_LANES(j, N0 << LM) Xs[j] = X[j]; }
      }
   }
   wv_sync();
}

/* denormalise_bands (bands.c:187), in place on one channel of X (downsample == 1): one lane per coefficient */
WV_DEV void denormalise_bands_wave(i32 *XF, const WV_LDS i32 *bandLogE, WV_LDS i32 *gains /* 2*21 ints */, int start, int end, int M, int silence, int downsample = 1)
{
   const int N = M * 120;
   int bound = M * ct_eBands[end];
   if (downsample != 1) bound = imin(bound, N / downsample);          /* nothing above the output Nyquist (bands.c:199) */
   if (silence) { bound = 0; start = end = 0; }
   wv_sync();
   FOR_LANES(i, NBE) {
      if (i >= start && i < end) {
         i32 lg = add32(bandLogE[i], shl32((i32)ct_eMeans[i], DB_SHIFT - 4));
         int shift = 17 - (lg >> DB_SHIFT);
         i32 g;
         if (shift >= 31) { shift = 0; g = 0; }
         else g = shl32(fx_exp2_db_frac(lg & ((1 << DB_SHIFT) - 1)), 2);
         if (shift < 0) { g = 2147483647; shift = 0; }
         gains[i] = g; gains[NBE + i] = shift;
      }
   }
   wv_sync();
   const int lm = ec_ilog((u32)M) - 1;
   FOR_LANES(j, N) {
      i32 v = 0;
      if (j >= M * ct_eBands[start] && j < bound) {
         const int bnd = ct_band_of[j >> lm];
         v = pshr32(mult32_32_q31(shl32(XF[j], 30 - NORM_SHIFT), gains[bnd]), gains[NBE + bnd]);
      }
      XF[j] = v;
   }
   wv_sync();
}
#endif
