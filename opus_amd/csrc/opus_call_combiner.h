/* opus_call_combiner.h — how the classic libopus entry points reach a machine that wants thousands of streams per launch.
 *
 * opus_encode() / opus_decode() work on ONE caller-owned state per call (include/opus.h:266, :516) and the reference lets any number of threads call them at the same
 * time on different states (include/opus.h:425-429).  A launch costs the same few milliseconds for one wave as for a few thousand, so calls that arrive while a launch
 * is in flight are not queued behind it one by one: they wait together, and the first caller to find the device free leads ONE launch for every waiting call of the
 * same shape (same kernel, rate, channels, frame size, byte budget), each with its own state record, input and output slot.  There is no timer and no added latency
 * for a lone caller (its group is itself); under T concurrent callers the groups settle at about T/2..T calls per launch.  The order of calls on one state is the
 * caller's (a state is in at most one call at a time, as the API requires); results do not depend on the grouping because streams never interact. */
#ifndef OPUS_AMD_CALL_COMBINER_H
#define OPUS_AMD_CALL_COMBINER_H
#include <mutex>
#include <condition_variable>
#include <deque>
#include <vector>
/* Req needs: bool done; int ret; bool same_launch(const Req &) const */
template <class Req> struct OaCallCombiner {
   std::mutex mu; std::condition_variable cv; std::deque<Req *> pending; bool busy = false;
   long long calls = 0, launches = 0;
   template <class Run> void submit(Req *rq, int cap, Run run)
   {
      std::unique_lock<std::mutex> lk(mu);
      calls++;
      pending.push_back(rq);
      while (!rq->done) {
         if (busy) { cv.wait(lk); continue; }
         busy = true;                                                    /* this caller leads the next launch: the oldest waiting call and every call that fits it */
         std::vector<Req *> grp;
         Req *head = pending.front();
         for (auto it = pending.begin(); it != pending.end() && (int)grp.size() < cap;) {
            if (*it == head || head->same_launch(**it)) { grp.push_back(*it); it = pending.erase(it); } else ++it;
         }
         launches++;
         lk.unlock();
         try { run(grp); } catch (...) { for (Req *g : grp) g->ret = -7 /* OPUS_ALLOC_FAIL */; }
         lk.lock();
         for (Req *g : grp) g->done = true;
         busy = false;
         cv.notify_all();
      }
   }
};
#endif
