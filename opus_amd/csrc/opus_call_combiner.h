/* opus_call_combiner.h — how the classic libopus entry points reach a machine that wants thousands of streams per launch.
 *
 * opus_encode() / opus_decode() work on ONE caller-owned state per call (include/opus.h:266, :516) and the reference lets any number of threads call them at the same
 * time on different states (include/opus.h:425-429).  A launch costs the same few milliseconds for one wave as for a few thousand, so calls that arrive while a launch
 * is in flight are not queued behind it one by one: they wait together, and the first caller to find the device free leads ONE launch for every waiting call of the
 * same shape (same kernel, rate, channels, frame size, byte budget), each with its own state record, input and output slot.
 *
 * Left alone, T steady callers settle into two alternating groups of T/2 (one group's launch runs while the other group's threads collect their results and come
 * back), and every caller waits for two launches per call.  The leader therefore lingers for the callers it has reason to expect: the THREADS that brought calls of its
 * own shape to the last two launches and that are not waiting yet -- until they arrive or `linger_us` (a small fraction of a launch) have passed.  A lone caller expects
 * nobody and never waits, however many states it cycles through (a mixer loop over N encoders is one caller: expecting its states instead of its thread would make every
 * one of its calls linger for a state that cannot arrive); a thread that stops calling is expected for two more launches, then forgotten.  The order of calls on one state is the caller's (a state is in at
 * most one call at a time, as the API requires); results do not depend on the grouping because streams never interact. */
#ifndef OPUS_AMD_CALL_COMBINER_H
#define OPUS_AMD_CALL_COMBINER_H
#include <mutex>
#include <condition_variable>
#include <deque>
#include <vector>
#include <algorithm>
#include <chrono>
#include <thread>
#include <functional>
/* Req needs: bool done; int ret; size_t tid (set here: the calling thread); const void *who() const (the state the call works on; never dereferenced here) and bool same_shape(const Req &) const, which compares
 * plain values only: the combiner keeps a COPY of the last launches' head requests, whose states may be gone by the time it looks at them */
template <class Req> struct OaCallCombiner {
   std::mutex mu; std::condition_variable cv, cv_lead; std::deque<Req *> pending; bool busy = false, lingering = false;
   std::vector<size_t> served[2];                                          /* calling threads of the last two launches, with the shape they were launched for */
   Req shape[2]; bool shape_set[2] = {false, false};
   long long calls = 0, launches = 0;
   static bool joins(const Req &head, const Req &r) { return &r == &head || (r.who() != head.who() && head.same_shape(r)); }
   size_t fits(const Req &head) const { size_t n = 0; for (const Req *r : pending) n += joins(head, *r); return n; }
   size_t expected(const Req &head) const                                  /* distinct recent calling threads of head's shape (head's own included) */
   {
      std::vector<size_t> ids;
      for (int k = 0; k < 2; k++) if (shape_set[k] && head.same_shape(shape[k]))
         ids.insert(ids.end(), served[k].begin(), served[k].end());
      ids.push_back(head.tid);
      std::sort(ids.begin(), ids.end());
      return (size_t)(std::unique(ids.begin(), ids.end()) - ids.begin());
   }
   template <class Run> void submit(Req *rq, int cap, int linger_us, Run run)
   {
      rq->tid = std::hash<std::thread::id>()(std::this_thread::get_id());
      std::unique_lock<std::mutex> lk(mu);
      calls++;
      pending.push_back(rq);
      if (lingering) cv_lead.notify_one();
      while (!rq->done) {
         if (busy) { cv.wait(lk); continue; }
         busy = true;                                                    /* this caller leads the next launch: the oldest waiting call and every call that fits it */
         Req *head = pending.front();
         if (linger_us > 0) {
            const size_t want = std::min(expected(*head), (size_t)cap);
            if (fits(*head) < want) {
               const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(linger_us);
               lingering = true;
               while (fits(*head) < want) if (cv_lead.wait_until(lk, deadline) == std::cv_status::timeout) break;
               lingering = false;
            }
         }
         std::vector<Req *> grp;
         for (auto it = pending.begin(); it != pending.end() && (int)grp.size() < cap;) {
            if (joins(*head, **it)) { grp.push_back(*it); it = pending.erase(it); } else ++it;
         }
         served[1].swap(served[0]); shape[1] = shape[0]; shape_set[1] = shape_set[0];
         served[0].clear(); for (Req *g : grp) served[0].push_back(g->tid);
         shape[0] = *head; shape_set[0] = true;
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
