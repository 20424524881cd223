/* oc_rangedec.c — 32-bit range decoder, mirror of oc_rangeenc.c.  TEST INFRASTRUCTURE (oracle), never shipped.
 * Oracle restatement of celt/entdec.c:91-246 (ec_dec_init :119, ec_decode :139, ec_dec_update :155, ec_dec_bit_logp
 * :164, ec_dec_icdf :179, ec_dec_uint :225, ec_dec_bits :249); constants celt/mfrngcod.h:35-47. */
#include "oc_celt.h"

#define SYM_BITS 8
#define SYM_MAX 255u
#define CODE_BITS 32
#define CODE_TOP 0x80000000u
#define CODE_BOT (CODE_TOP >> SYM_BITS)
#define CODE_EXTRA ((CODE_BITS - 2) % SYM_BITS + 1)      /* 7 */
#define UINT_BITS 8
#define WINDOW_SIZE 32

static int read_front(oc_ec *d) { return d->offs < d->storage ? d->buf[d->offs++] : 0; }
static int read_back(oc_ec *d) { return d->end_offs < d->storage ? d->buf[d->storage - ++(d->end_offs)] : 0; }

/* entdec.c:102 */
static void dec_normalize(oc_ec *d)
{
   while (d->rng <= CODE_BOT) {
      int sym;
      d->nbits_total += SYM_BITS;
      d->rng <<= SYM_BITS;
      sym = d->rem;
      d->rem = read_front(d);
      sym = (sym << SYM_BITS | d->rem) >> (SYM_BITS - CODE_EXTRA);
      d->val = ((d->val << SYM_BITS) + (SYM_MAX & ~sym)) & (CODE_TOP - 1);
   }
}
void oc_ec_dec_init(oc_ec *d, const u8 *buf, u32 storage)
{
   d->buf = (u8 *)buf; d->storage = storage; d->end_offs = 0; d->end_window = 0; d->nend_bits = 0;
   d->nbits_total = CODE_BITS + 1 - ((CODE_BITS - CODE_EXTRA) / SYM_BITS) * SYM_BITS;
   d->offs = 0;
   d->rng = 1U << CODE_EXTRA;
   d->rem = read_front(d);
   d->val = d->rng - 1 - (d->rem >> (SYM_BITS - CODE_EXTRA));
   d->error = 0; d->ext = 0;
   dec_normalize(d);
}
unsigned oc_ec_decode(oc_ec *d, unsigned ft)
{
   d->ext = d->rng / ft;
   unsigned s = (unsigned)(d->val / d->ext);
   return ft - (s + 1 < ft ? s + 1 : ft);
}
unsigned oc_ec_decode_bin(oc_ec *d, unsigned bits)
{
   d->ext = d->rng >> bits;
   unsigned s = (unsigned)(d->val / d->ext);
   return (1U << bits) - (s + 1U < (1U << bits) ? s + 1U : (1U << bits));
}
void oc_ec_dec_update(oc_ec *d, unsigned fl, unsigned fh, unsigned ft)
{
   u32 s = d->ext * (ft - fh);
   d->val -= s;
   d->rng = fl > 0 ? d->ext * (fh - fl) : d->rng - s;
   dec_normalize(d);
}
int oc_ec_dec_bit_logp(oc_ec *d, unsigned logp)
{
   u32 r = d->rng, v = d->val, s = r >> logp;
   int ret = v < s;
   if (!ret) d->val = v - s;
   d->rng = ret ? s : r - s;
   dec_normalize(d);
   return ret;
}
int oc_ec_dec_icdf(oc_ec *d, const u8 *icdf, unsigned ftb)
{
   u32 s = d->rng, v = d->val, r = s >> ftb, t;
   int ret = -1;
   do { t = s; s = r * icdf[++ret]; } while (v < s);
   d->val = v - s;
   d->rng = t - s;
   dec_normalize(d);
   return ret;
}
u32 oc_ec_dec_bits(oc_ec *d, unsigned bits)
{
   u32 window = d->end_window;
   int available = d->nend_bits;
   if ((unsigned)available < bits) {
      do { window |= (u32)read_back(d) << available; available += SYM_BITS; } while (available <= WINDOW_SIZE - SYM_BITS);
   }
   u32 ret = window & (((u32)1 << bits) - 1U);
   window >>= bits;
   available -= bits;
   d->end_window = window; d->nend_bits = available; d->nbits_total += bits;
   return ret;
}
u32 oc_ec_dec_uint(oc_ec *d, u32 ft_)
{
   unsigned ft, s;
   int ftb;
   ft_--;
   ftb = ec_ilog(ft_);
   if (ftb > UINT_BITS) {
      u32 t;
      ftb -= UINT_BITS;
      ft = (unsigned)(ft_ >> ftb) + 1;
      s = oc_ec_decode(d, ft);
      oc_ec_dec_update(d, s, s + 1, ft);
      t = (u32)s << ftb | oc_ec_dec_bits(d, ftb);
      if (t <= ft_) return t;
      d->error = 1;
      return ft_;
   } else {
      ft_++;
      s = oc_ec_decode(d, (unsigned)ft_);
      oc_ec_dec_update(d, s, s + 1, (unsigned)ft_);
      return s;
   }
}
