/* celt_ecdec.h — range decoder on the register-resident EcCtx (lane-0 serial code), mirror of celt_ec.h.
 * Semantics: celt/entdec.c:91-266.  The frame's bytes sit in LDS (L->packet + 1), loaded once, coalesced -- or, for the lane = stream SILK decoder
 * (silk_dec_lane.h), in the stream's packet slot in HBM: like the encoder's symbol coder the functions are generic over the byte pointer (ECB). */
#ifndef OPUS_AMD_CELT_ECDEC_H
#define OPUS_AMD_CELT_ECDEC_H
#define DEC_CODE_EXTRA 7
template <class ECB> WV_DEV int ecd_read_front(EC_ARGS_G) { return e->offs < e->storage ? buf[e->offs++] : 0; }
template <class ECB> WV_DEV int ecd_read_back(EC_ARGS_G) { return e->end_offs < e->storage ? buf[e->storage - ++(e->end_offs)] : 0; }
template <class ECB> WV_DEV void ecd_normalize(EC_ARGS_G)
{
   while (e->rng <= CODE_BOT) {
      int sym;
      e->nbits_total += SYM_BITS;
      e->rng <<= SYM_BITS;
      sym = e->rem;
      e->rem = ecd_read_front(EC_PASS);
      sym = (sym << SYM_BITS | e->rem) >> (SYM_BITS - DEC_CODE_EXTRA);
      e->val = ((e->val << SYM_BITS) + (SYM_MAX & ~sym)) & (CODE_TOP - 1);
   }
}
template <class ECB> WV_DEV void k_ec_dec_init(EC_ARGS_G, u32 storage)
{
   e->storage = storage; e->end_offs = 0; e->end_window = 0; e->nend_bits = 0;
   e->nbits_total = 32 + 1 - ((32 - DEC_CODE_EXTRA) / SYM_BITS) * SYM_BITS;
   e->offs = 0;
   e->rng = 1U << DEC_CODE_EXTRA;
   e->rem = ecd_read_front(EC_PASS);
   e->val = e->rng - 1 - (e->rem >> (SYM_BITS - DEC_CODE_EXTRA));
   e->error = 0; e->ext = 0;
   ecd_normalize(EC_PASS);
}
template <class ECB> WV_DEV unsigned k_ec_decode(EC_ARGS_G, unsigned ft)
{
   e->ext = e->rng / ft;
   unsigned s = (unsigned)(e->val / e->ext);
   return ft - (s + 1 < ft ? s + 1 : ft);
}
template <class ECB> WV_DEV unsigned k_ec_decode_bin(EC_ARGS_G, unsigned bits)
{
   e->ext = e->rng >> bits;
   unsigned s = (unsigned)(e->val / e->ext);
   return (1U << bits) - (s + 1U < (1U << bits) ? s + 1U : (1U << bits));
}
template <class ECB> WV_DEV void k_ec_dec_update(EC_ARGS_G, unsigned fl, unsigned fh, unsigned ft)
{
   u32 s = e->ext * (ft - fh);
   e->val -= s;
   e->rng = fl > 0 ? e->ext * (fh - fl) : e->rng - s;
   ecd_normalize(EC_PASS);
}
template <class ECB> WV_DEV int k_ec_dec_bit_logp(EC_ARGS_G, unsigned logp)
{
   u32 r = e->rng, v = e->val, s = r >> logp;
   int ret = v < s;
   if (!ret) e->val = v - s;
   e->rng = ret ? s : r - s;
   ecd_normalize(EC_PASS);
   return ret;
}
template <class ECB> WV_DEV int k_ec_dec_icdf(EC_ARGS_G, const u8 *icdf, unsigned ftb)
{
   u32 s = e->rng, v = e->val, r = s >> ftb, t;
   int ret = -1;
   do { t = s; s = r * icdf[++ret]; } while (v < s);
   e->val = v - s;
   e->rng = t - s;
   ecd_normalize(EC_PASS);
   return ret;
}
template <class ECB> WV_DEV u32 k_ec_dec_bits(EC_ARGS_G, unsigned bits)
{
   u32 window = e->end_window;
   int available = e->nend_bits;
   if ((unsigned)available < bits) {
      do { window |= (u32)ecd_read_back(EC_PASS) << available; available += SYM_BITS; } while (available <= 32 - SYM_BITS);
   }
   u32 ret = window & (((u32)1 << bits) - 1U);
   window >>= bits;
   available -= bits;
   e->end_window = window; e->nend_bits = available; e->nbits_total += bits;
   return ret;
}
template <class ECB> WV_DEV u32 k_ec_dec_uint(EC_ARGS_G, u32 ft_)
{
   unsigned ft, s;
   int ftb;
   ft_--;
   ftb = ec_ilog(ft_);
   if (ftb > 8) {
      u32 t;
      ftb -= 8;
      ft = (unsigned)(ft_ >> ftb) + 1;
      s = k_ec_decode(EC_PASS, ft);
      k_ec_dec_update(EC_PASS, s, s + 1, ft);
      t = (u32)s << ftb | k_ec_dec_bits(EC_PASS, ftb);
      if (t <= ft_) return t;
      e->error = 1;
      return ft_;
   } else {
      ft_++;
      s = k_ec_decode(EC_PASS, (unsigned)ft_);
      k_ec_dec_update(EC_PASS, s, s + 1, (unsigned)ft_);
      return s;
   }
}
#endif
