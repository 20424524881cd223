/* ref_expose/x_tables.c — TEST INFRASTRUCTURE.  Compiled against the reference tree in place
 * (never copied) to dump the normative constant tables of the standard 48 kHz mode and of the
 * energy/PVQ coders, so tools/gen_tables.py can emit them in this repo's own layout and the
 * tests can check the committed tables against the reference build. */
#include <stdio.h>
#include "quant_bands.c"   /* e_prob_model, eMeans, pred/beta coefs (celt/quant_bands.c:44-140) */
#include "modes.h"
#define DUMP(name, ptr, n, fmt, cast) do { fprintf(f, "%s %d\n", name, (int)(n)); \
   for (int i_ = 0; i_ < (int)(n); i_++) fprintf(f, fmt " ", (cast)(ptr)[i_]); fprintf(f, "\n"); } while (0)

extern const opus_uint32 *ref_pvq_u_data(int *n);
extern const int *ref_log2_frac_table(int *n);

int ref_dump_tables(const char *path)
{
   int err;
   const CELTMode *m = opus_custom_mode_create(48000, 960, &err);
   FILE *f = fopen(path, "w");
   if (!m || !f) return -1;
   DUMP("eBands", m->eBands, m->nbEBands + 1, "%d", int);
   DUMP("allocVectors", m->allocVectors, m->nbAllocVectors * m->nbEBands, "%d", int);
   DUMP("logN", m->logN, m->nbEBands, "%d", int);
   DUMP("window", m->window, m->overlap, "%d", int);
   DUMP("preemph", m->preemph, 4, "%d", int);
   DUMP("mdct_trig", m->mdct.trig, 960 + 480 + 240 + 120, "%d", int);
   DUMP("cache_index", m->cache.index, 105, "%d", int);
   DUMP("cache_bits", m->cache.bits, m->cache.size, "%d", int);
   DUMP("cache_caps", m->cache.caps, 168, "%d", int);
   for (int k = 0; k < 4; k++) {
      const kiss_fft_state *st = m->mdct.kfft[k];
      char nm[64];
      sprintf(nm, "fft%d_bitrev", k); DUMP(nm, st->bitrev, st->nfft, "%d", int);
      sprintf(nm, "fft%d_factors", k); DUMP(nm, st->factors, 16, "%d", int);
      int misc[4] = {st->nfft, st->scale, st->scale_shift, st->shift};
      sprintf(nm, "fft%d_misc", k); DUMP(nm, misc, 4, "%d", int);
   }
   {
      const kiss_fft_state *st = m->mdct.kfft[0];
      fprintf(f, "fft_twiddles %d\n", 2 * st->nfft);
      for (int i = 0; i < st->nfft; i++) fprintf(f, "%d %d ", st->twiddles[i].r, st->twiddles[i].i);
      fprintf(f, "\n");
   }
   DUMP("eMeans", eMeans, 25, "%d", int);
   DUMP("e_prob_model", &e_prob_model[0][0][0], 4 * 2 * 42, "%d", int);
   { int n; const opus_uint32 *u = ref_pvq_u_data(&n); DUMP("pvq_u_data", u, n, "%u", unsigned); }
   fclose(f);
   return 0;
}
