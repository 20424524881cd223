/* oc_rangeenc.c — 32-bit range encoder with raw bits packed from the buffer tail.
 * Oracle restatement of celt/entenc.c:60-305, celt/entcode.c:69-92 (ec_tell_frac),
 * constants celt/mfrngcod.h:35-47 (SYM_BITS 8, CODE_BITS 32, CODE_SHIFT 23). */
#include "oc_celt.h"

#define SYM_BITS 8
#define SYM_MAX 255u
#define CODE_SHIFT 23
#define CODE_TOP 0x80000000u
#define CODE_BOT (CODE_TOP >> SYM_BITS)

static int put_front(oc_ec *e, unsigned v)
{
   if (e->offs + e->end_offs >= e->storage) return -1;
   e->buf[e->offs++] = (u8)v;
   return 0;
}
static int put_back(oc_ec *e, unsigned v)
{
   if (e->offs + e->end_offs >= e->storage) return -1;
   e->buf[e->storage - ++(e->end_offs)] = (u8)v;
   return 0;
}
/* carry propagation: entenc.c:86 */
static void carry_out(oc_ec *e, int c)
{
   if (c != (int)SYM_MAX) {
      int carry = c >> SYM_BITS;
      if (e->rem >= 0) e->error |= put_front(e, e->rem + carry);
      if (e->ext > 0) {
         unsigned sym = (SYM_MAX + carry) & SYM_MAX;
         do e->error |= put_front(e, sym); while (--(e->ext) > 0);
      }
      e->rem = c & SYM_MAX;
   } else e->ext++;
}
static void renorm(oc_ec *e)
{
   while (e->rng <= CODE_BOT) {
      carry_out(e, (int)(e->val >> CODE_SHIFT));
      e->val = (e->val << SYM_BITS) & (CODE_TOP - 1);
      e->rng <<= SYM_BITS;
      e->nbits_total += SYM_BITS;
   }
}
void oc_ec_enc_init(oc_ec *e, u8 *buf, u32 size)
{
   e->buf = buf; e->end_offs = 0; e->end_window = 0; e->nend_bits = 0;
   e->nbits_total = 33; e->offs = 0; e->rng = CODE_TOP; e->rem = -1; e->val = 0; e->ext = 0;
   e->storage = size; e->error = 0;
}
int oc_ec_tell(const oc_ec *e) { return e->nbits_total - ec_ilog(e->rng); }
u32 oc_ec_tell_frac(const oc_ec *e)
{
   static const unsigned correction[8] = {35733, 38967, 42495, 46340, 50535, 55109, 60097, 65535};
   u32 nbits = (u32)e->nbits_total << BITRES;
   int l = ec_ilog(e->rng);
   u32 r = e->rng >> (l - 16);
   unsigned b = (r >> 12) - 8;
   b += r > correction[b];
   l = (l << 3) + b;
   return nbits - l;
}
void oc_ec_encode(oc_ec *e, unsigned fl, unsigned fh, unsigned ft)
{
   u32 r = e->rng / ft;
   if (fl > 0) { e->val += e->rng - r * (ft - fl); e->rng = r * (fh - fl); }
   else e->rng -= r * (ft - fh);
   renorm(e);
}
void oc_ec_encode_bin(oc_ec *e, unsigned fl, unsigned fh, unsigned bits)
{
   u32 r = e->rng >> bits;
   if (fl > 0) { e->val += e->rng - r * ((1U << bits) - fl); e->rng = r * (fh - fl); }
   else e->rng -= r * ((1U << bits) - fh);
   renorm(e);
}
void oc_ec_enc_bit_logp(oc_ec *e, int val, unsigned logp)
{
   u32 r = e->rng, l = e->val, s = r >> logp;
   r -= s;
   if (val) e->val = l + r;
   e->rng = val ? s : r;
   renorm(e);
}
void oc_ec_enc_icdf(oc_ec *e, int s, const u8 *icdf, unsigned ftb)
{
   u32 r = e->rng >> ftb;
   if (s > 0) { e->val += e->rng - r * icdf[s - 1]; e->rng = r * (u32)(icdf[s - 1] - icdf[s]); }
   else e->rng -= r * icdf[s];
   renorm(e);
}
void oc_ec_enc_bits(oc_ec *e, u32 fl, unsigned bits)
{
   u32 window = e->end_window;
   int used = e->nend_bits;
   if (used + bits > 32) {
      do { e->error |= put_back(e, window & SYM_MAX); window >>= SYM_BITS; used -= SYM_BITS; } while (used >= SYM_BITS);
   }
   window |= fl << used;
   used += bits;
   e->end_window = window; e->nend_bits = used; e->nbits_total += bits;
}
void oc_ec_enc_uint(oc_ec *e, u32 fl, u32 ft)
{
   ft--;
   int ftb = ec_ilog(ft);
   if (ftb > 8) {
      ftb -= 8;
      unsigned t = (ft >> ftb) + 1, l = (unsigned)(fl >> ftb);
      oc_ec_encode(e, l, l + 1, t);
      oc_ec_enc_bits(e, fl & (((u32)1 << ftb) - 1U), ftb);
   } else oc_ec_encode(e, fl, fl + 1, ft + 1);
}
void oc_ec_enc_patch_initial_bits(oc_ec *e, unsigned val, unsigned nbits)
{
   int shift = SYM_BITS - nbits;
   unsigned mask = ((1 << nbits) - 1) << shift;
   if (e->offs > 0) e->buf[0] = (u8)((e->buf[0] & ~mask) | val << shift);
   else if (e->rem >= 0) e->rem = (e->rem & ~mask) | val << shift;
   else if (e->rng <= (CODE_TOP >> nbits))
      e->val = (e->val & ~((u32)mask << CODE_SHIFT)) | (u32)val << (CODE_SHIFT + shift);
   else e->error = -1;
}
void oc_ec_enc_shrink(oc_ec *e, u32 size)
{
   memmove(e->buf + size - e->end_offs, e->buf + e->storage - e->end_offs, e->end_offs);
   e->storage = size;
}
void oc_ec_enc_done(oc_ec *e)
{
   int l = 32 - ec_ilog(e->rng);
   u32 msk = (CODE_TOP - 1) >> l;
   u32 end = (e->val + msk) & ~msk;
   if ((end | msk) >= e->val + e->rng) { l++; msk >>= 1; end = (e->val + msk) & ~msk; }
   while (l > 0) {
      carry_out(e, (int)(end >> CODE_SHIFT));
      end = (end << SYM_BITS) & (CODE_TOP - 1);
      l -= SYM_BITS;
   }
   if (e->rem >= 0 || e->ext > 0) carry_out(e, 0);
   u32 window = e->end_window;
   int used = e->nend_bits;
   while (used >= SYM_BITS) { e->error |= put_back(e, window & SYM_MAX); window >>= SYM_BITS; used -= SYM_BITS; }
   if (!e->error) {
      memset(e->buf + e->offs, 0, e->storage - e->offs - e->end_offs);
      if (used > 0) {
         if (e->end_offs >= e->storage) e->error = -1;
         else {
            l = -l;
            if (e->offs + e->end_offs >= e->storage && l < used) { window &= (1 << l) - 1; e->error = -1; }
            e->buf[e->storage - e->end_offs - 1] |= (u8)window;
         }
      }
   }
}
