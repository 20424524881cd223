/* celt_ec.h — range encoder operating on an LDS-resident context and packet buffer (lane-0 serial code).
 * Semantics: celt/entenc.c:60-305, celt/entcode.c:69-92 (ec_tell_frac), celt/mfrngcod.h:35-47.  The whole
 * packet (<=1275 B) stays in LDS until the final coalesced store, so the snapshot/rollback the encoder
 * needs (two-pass coarse energy, theta-RDO) is an LDS-to-LDS copy. */
#ifndef OPUS_AMD_CELT_EC_H
#define OPUS_AMD_CELT_EC_H
struct EcCtx { u32 storage, end_offs, end_window; i32 nend_bits, nbits_total; u32 offs, rng, val, ext; i32 rem, error; };
/* The coder state is worked on in registers (a private EcCtx) inside lane-0 sections and parked in LDS between them:
 * EC_BEGIN loads the 11 words with back-to-back ds_reads, EC_END stores them; every symbol in between is pure VALU
 * plus the byte stores into the LDS packet. */
#define EC_ARGS EcCtx *e, WV_LDS u8 *buf
#define EC_PASS e, buf
/* the symbol coder itself is generic over where the bytes go (ECB: any pointer to u8): the SILK quantiser kernel codes 16 streams per wave, each lane into its own
 * stream's buffer in HBM (opus_sh_split.h) */
#define EC_ARGS_G EcCtx *e, ECB buf
WV_DEV void ec_ld(EcCtx *d, const WV_LDS EcCtx *s)
{
   d->storage = s->storage; d->end_offs = s->end_offs; d->end_window = s->end_window; d->nend_bits = s->nend_bits; d->nbits_total = s->nbits_total;
   d->offs = s->offs; d->rng = s->rng; d->val = s->val; d->ext = s->ext; d->rem = s->rem; d->error = s->error;
}
WV_DEV void ec_st(WV_LDS EcCtx *d, const EcCtx *s)
{
   d->storage = s->storage; d->end_offs = s->end_offs; d->end_window = s->end_window; d->nend_bits = s->nend_bits; d->nbits_total = s->nbits_total;
   d->offs = s->offs; d->rng = s->rng; d->val = s->val; d->ext = s->ext; d->rem = s->rem; d->error = s->error;
}
WV_DEV void ec_cp_lds(WV_LDS EcCtx *d, const WV_LDS EcCtx *s) { EcCtx t; ec_ld(&t, s); ec_st(d, &t); }
#define EC_BEGIN EcCtx ec_; ec_ld(&ec_, &L->ec); EcCtx *e = &ec_; WV_LDS u8 *buf = L->packet + 1
#define EC_END ec_st(&L->ec, &ec_)
#define SYM_BITS 8
#define SYM_MAX 255u
#define CODE_SHIFT 23
#define CODE_TOP 0x80000000u
#define CODE_BOT (CODE_TOP >> SYM_BITS)

template <class ECB> WV_DEV int ec_put_front(EC_ARGS_G, unsigned v)
{
   if (e->offs + e->end_offs >= e->storage) return -1;
   buf[e->offs++] = (u8)v;
   return 0;
}
template <class ECB> WV_DEV int ec_put_back(EC_ARGS_G, unsigned v)
{
   if (e->offs + e->end_offs >= e->storage) return -1;
   buf[e->storage - ++(e->end_offs)] = (u8)v;
   return 0;
}
/* carry propagation: entenc.c:86 */
template <class ECB> WV_DEV void ec_carry_out(EC_ARGS_G, int c)
{
   if (c != (int)SYM_MAX) {
      int carry = c >> SYM_BITS;
      if (e->rem >= 0) e->error |= ec_put_front(EC_PASS, e->rem + carry);
      if (e->ext > 0) {
         unsigned sym = (SYM_MAX + carry) & SYM_MAX;
         do e->error |= ec_put_front(EC_PASS, sym); while (--(e->ext) > 0);
      }
      e->rem = c & SYM_MAX;
   } else e->ext++;
}
template <class ECB> WV_DEV void ec_renorm(EC_ARGS_G)
{
   while (e->rng <= CODE_BOT) {
      ec_carry_out(EC_PASS, (int)(e->val >> CODE_SHIFT));
      e->val = (e->val << SYM_BITS) & (CODE_TOP - 1);
      e->rng <<= SYM_BITS;
      e->nbits_total += SYM_BITS;
   }
}
WV_DEV void k_ec_enc_init(EC_ARGS, u32 size)
{
   e->end_offs = 0; e->end_window = 0; e->nend_bits = 0;
   e->nbits_total = 33; e->offs = 0; e->rng = CODE_TOP; e->rem = -1; e->val = 0; e->ext = 0;
   e->storage = size; e->error = 0;
}
template <class ECB> WV_DEV int k_ec_tell(const EcCtx *e, ECB buf) { (void)buf; return e->nbits_total - ec_ilog(e->rng); }
WV_DEV u32 k_ec_tell_frac(EC_ARGS)
{
   const unsigned correction[8] = {35733, 38967, 42495, 46340, 50535, 55109, 60097, 65535};
   u32 nbits = (u32)e->nbits_total << BITRES;
   int l = ec_ilog(e->rng);
   u32 r = e->rng >> (l - 16);
   unsigned b = (r >> 12) - 8;
   b += r > correction[b];
   l = (l << 3) + b;
   return nbits - l;
}
WV_DEV int ec_tell_lds(const WV_LDS EcCtx *e) { return e->nbits_total - ec_ilog(e->rng); }
WV_DEV u32 ec_tell_frac_lds(const WV_LDS EcCtx *e)
{
   const unsigned correction[8] = {35733, 38967, 42495, 46340, 50535, 55109, 60097, 65535};
   u32 rng = e->rng, nbits = (u32)e->nbits_total << BITRES;
   int l = ec_ilog(rng);
   u32 r = rng >> (l - 16);
   unsigned b = (r >> 12) - 8;
   b += r > correction[b];
   l = (l << 3) + b;
   return nbits - l;
}
template <class ECB> WV_DEV void k_ec_encode(EC_ARGS_G, unsigned fl, unsigned fh, unsigned ft)
{
   u32 r = e->rng / ft;
   if (fl > 0) { e->val += e->rng - r * (ft - fl); e->rng = r * (fh - fl); }
   else e->rng -= r * (ft - fh);
   ec_renorm(EC_PASS);
}
WV_DEV void k_ec_encode_bin(EC_ARGS, unsigned fl, unsigned fh, unsigned bits)
{
   u32 r = e->rng >> bits;
   if (fl > 0) { e->val += e->rng - r * ((1U << bits) - fl); e->rng = r * (fh - fl); }
   else e->rng -= r * ((1U << bits) - fh);
   ec_renorm(EC_PASS);
}
template <class ECB> WV_DEV void k_ec_enc_bit_logp(EC_ARGS_G, int val, unsigned logp)
{
   u32 r = e->rng, l = e->val, s = r >> logp;
   r -= s;
   if (val) e->val = l + r;
   e->rng = val ? s : r;
   ec_renorm(EC_PASS);
}
template <class ECB> WV_DEV void k_ec_enc_icdf(EC_ARGS_G, int s, const u8 *icdf, unsigned ftb)
{
   u32 r = e->rng >> ftb;
   if (s > 0) { e->val += e->rng - r * icdf[s - 1]; e->rng = r * (u32)(icdf[s - 1] - icdf[s]); }
   else e->rng -= r * icdf[s];
   ec_renorm(EC_PASS);
}
template <class ECB> WV_DEV void k_ec_enc_bits(EC_ARGS_G, u32 fl, unsigned bits)
{
   u32 window = e->end_window;
   int used = e->nend_bits;
   if (used + bits > 32) {
      do { e->error |= ec_put_back(EC_PASS, window & SYM_MAX); window >>= SYM_BITS; used -= SYM_BITS; } while (used >= SYM_BITS);
   }
   window |= fl << used;
   used += bits;
   e->end_window = window; e->nend_bits = used; e->nbits_total += bits;
}
template <class ECB> WV_DEV void k_ec_enc_uint(EC_ARGS_G, u32 fl, u32 ft)
{
   ft--;
   int ftb = ec_ilog(ft);
   if (ftb > 8) {
      ftb -= 8;
      unsigned t = (ft >> ftb) + 1, l = (unsigned)(fl >> ftb);
      k_ec_encode(EC_PASS, l, l + 1, t);
      k_ec_enc_bits(EC_PASS, fl & (((u32)1 << ftb) - 1U), ftb);
   } else k_ec_encode(EC_PASS, fl, fl + 1, ft + 1);
}
template <class ECB> WV_DEV void k_ec_enc_patch_initial_bits(EC_ARGS_G, unsigned val, unsigned nbits)
{
   int shift = SYM_BITS - nbits;
   unsigned mask = ((1 << nbits) - 1) << shift;
   if (e->offs > 0) buf[0] = (u8)((buf[0] & ~mask) | val << shift);
   else if (e->rem >= 0) e->rem = (e->rem & ~mask) | val << shift;
   else if (e->rng <= (CODE_TOP >> nbits))
      e->val = (e->val & ~((u32)mask << CODE_SHIFT)) | (u32)val << (CODE_SHIFT + shift);
   else e->error = -1;
}
WV_DEV void k_ec_enc_shrink(EC_ARGS, u32 size)
{
   if (size < e->storage) for (u32 i_ = 0; i_ < e->end_offs; i_++) buf[size - e->end_offs + i_] = buf[e->storage - e->end_offs + i_];
   else for (u32 i_ = e->end_offs; i_-- > 0;) buf[size - e->end_offs + i_] = buf[e->storage - e->end_offs + i_];
   e->storage = size;
}
WV_DEV void k_ec_enc_done(EC_ARGS)
{
   int l = 32 - ec_ilog(e->rng);
   u32 msk = (CODE_TOP - 1) >> l;
   u32 end = (e->val + msk) & ~msk;
   if ((end | msk) >= e->val + e->rng) { l++; msk >>= 1; end = (e->val + msk) & ~msk; }
   while (l > 0) {
      ec_carry_out(EC_PASS, (int)(end >> CODE_SHIFT));
      end = (end << SYM_BITS) & (CODE_TOP - 1);
      l -= SYM_BITS;
   }
   if (e->rem >= 0 || e->ext > 0) ec_carry_out(EC_PASS, 0);
   u32 window = e->end_window;
   int used = e->nend_bits;
   while (used >= SYM_BITS) { e->error |= ec_put_back(EC_PASS, window & SYM_MAX); window >>= SYM_BITS; used -= SYM_BITS; }
   if (!e->error) {
      for (u32 i_ = e->offs; i_ < e->storage - e->end_offs; i_++) buf[i_] = 0;
      if (used > 0) {
         if (e->end_offs >= e->storage) e->error = -1;
         else {
            l = -l;
            if (e->offs + e->end_offs >= e->storage && l < used) { window &= (1 << l) - 1; e->error = -1; }
            buf[e->storage - e->end_offs - 1] |= (u8)window;
         }
      }
   }
}

#endif
