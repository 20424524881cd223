/* tools/classic_threads_bench.cpp — T native threads, each with its own OpusEncoder / OpusDecoder, calling opus_encode() / opus_decode() in a loop against
 * opus_amd/libopus_amd.so (BASELINE config 2 settings): calls per second and calls per launch.  The Python twin (tools/classic_threads_bench.py) pays for the
 * interpreter lock when hundreds of threads collect results at once; this one shows what a C caller gets.
 *   g++ -O2 -std=c++17 -pthread tools/classic_threads_bench.cpp -Iinclude -Lopus_amd -lopus_amd -Wl,-rpath,$PWD/opus_amd -o gpurun_out/ctb && gpurun_out/ctb [frames] */
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>
#include <chrono>
#include "opus_amd.h"
extern "C" { typedef struct OpusDecoder OpusDecoder; }
static void synth(std::vector<opus_int16> &x, int n, unsigned seed)
{
   x.resize((size_t)n * 2);
   double f0 = 80 + (seed * 37 % 320), ph = 0; unsigned r = seed * 2654435761u + 1;
   for (int i = 0; i < n; i++) {
      ph += 2 * M_PI * f0 / 48000.0; double s = 0;
      for (int k = 1; k < 12; k++) s += sin(k * ph) / k;
      r = r * 1664525u + 1013904223u; const double noise = ((int)(r >> 16) - 32768) / 32768.0 * 0.05;
      const double gate = sin(2 * M_PI * 2 * i / 48000.0) > -0.3;
      const int v = (int)(9000 * (s * gate / 2 + noise));
      x[2 * i] = (opus_int16)v; x[2 * i + 1] = (opus_int16)(v * 3 / 4);
   }
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv)
{
   const int nf = argc > 1 ? atoi(argv[1]) : 50;
   const int Ts[] = {1, 4, 16, 64, 256, 1024};
   for (int decode = 0; decode < 2; decode++)
   for (int T : Ts) {
      std::vector<OpusEncoder *> enc((size_t)T); std::vector<OpusDecoder *> dec((size_t)T);
      std::vector<std::vector<opus_int16>> x(16);
      for (int k = 0; k < 16; k++) synth(x[k], nf * 960, 7 + k);
      int err;
      for (int k = 0; k < T; k++) {
         enc[k] = opus_encoder_create(48000, 2, 2051, &err);
         opus_encoder_ctl(enc[k], OPUS_SET_BITRATE_REQUEST, 128000); opus_encoder_ctl(enc[k], OPUS_SET_COMPLEXITY_REQUEST, 10);
         dec[k] = opus_decoder_create(48000, 2, &err);
      }
      std::vector<std::vector<unsigned char>> pk((size_t)nf); std::vector<int> pl((size_t)nf);
      if (decode) {
         OpusEncoder *e = opus_encoder_create(48000, 2, 2051, &err); opus_encoder_ctl(e, OPUS_SET_BITRATE_REQUEST, 128000);
         for (int i = 0; i < nf; i++) { pk[i].resize(1276); pl[i] = opus_encode(e, x[0].data() + (size_t)i * 1920, 960, pk[i].data(), 1276); }
         opus_encoder_destroy(e);
      }
      { unsigned char b[1276]; opus_int16 o[1920]; opus_encode(enc[0], x[0].data(), 960, b, 1276); if (decode) opus_decode(dec[0], pk[0].data(), pl[0], o, 960, 0); }   /* device arrays of the shape */
      long long s0[4], s1[4]; opusgpu_classic_call_stats(s0);
      const double t0 = now();
      std::vector<std::thread> th;
      for (int k = 0; k < T; k++) th.emplace_back([&, k]() {
         unsigned char b[1276]; opus_int16 o[1920];
         for (int i = 0; i < nf; i++) {
            if (decode) { if (opus_decode(dec[k], pk[i].data(), pl[i], o, 960, 0) != 960) abort(); }
            else if (opus_encode(enc[k], x[k % 16].data() + (size_t)i * 1920, 960, b, 1276) <= 0) abort();
         }
      });
      for (auto &t : th) t.join();
      const double dt = now() - t0; opusgpu_classic_call_stats(s1);
      const long long c = s1[decode * 2] - s0[decode * 2], l = s1[decode * 2 + 1] - s0[decode * 2 + 1];
      printf("{\"threads\": %d, \"op\": \"%s\", \"calls_per_s\": %.1f, \"calls_per_launch\": %.2f, \"ms_per_call_seen_by_a_thread\": %.2f}\n", T, decode ? "opus_decode" : "opus_encode",
             c / dt, (double)c / (l ? l : 1), 1e3 * dt / nf);
      fflush(stdout);
      for (int k = 0; k < T; k++) { opus_encoder_destroy(enc[k]); opus_decoder_destroy(dec[k]); }
   }
   return 0;
}
