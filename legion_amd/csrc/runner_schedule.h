// runner_schedule.h -- which launch group goes into which pipeline slot, and WHEN a slot's lanes may be overwritten.
//
// Host-only, no HIP: GPURunner (server.hip) drives it, tests/cpu/runner_schedule_test.cpp simulates a trainer against it.
//
// The wire protocol (SS/engine/ipc_service.cu:283-291, TB/ipc_cuda_kernel.cu:97-106) gives the server ONE piece of knowledge about the
// trainer: semaphore tokens.  The trainer posts sem_r once per pipe slot at start-up (two tokens) and once per synchronize();
// the server consumes one token per batch, in order.  So when the server has consumed the token of batch k, the trainer has
// called synchronize() at least k-1 times: batches 0 .. k-2 are RELEASED (it will never touch their memory again), batch k-1
// may still be in use, batch k is about to be handed over.
//
// With the `views` hand-over a batch is read in place -- in the lane its launch group wrote it into -- so a pipeline slot's
// lanes may be overwritten (= the next group submitted into that slot) only when every batch of the group that used the slot
// last is released: a group that ended at batch e (exclusive) is out of the way once e - 1 <= k - 2.  The `gather` and `copy`
// hand-overs satisfy the same rule a fortiori (a batch is posted only after its hand-over kernel has read the lane).
#pragma once

#include <cstdint>
#include <deque>
#include <vector>

struct RunnerSchedule {
    struct Group { int slot; int32_t first, n; bool complete; };

    int slots = 3;
    std::deque<Group> groups;          // submitted, not yet fully handed over; front = the one being handed over
    std::vector<int32_t> slot_end;     // [slot] exclusive end of the group that last used it, -1: never used
    int32_t next_first = 0;            // first batch of the next group to submit
    int submit_count = 0;              // groups submitted so far: the pipeline hands its slots out round robin

    void reset(int slots_)
    {
        slots = slots_;
        groups.clear();
        slot_end.assign(slots, -1);
        next_first = 0;
        submit_count = 0;
    }
    int next_slot() const { return submit_count % slots; }
    // with the token of batch k consumed (k < 0: before any token): may the next group be submitted now?
    bool may_submit(int32_t k) const
    {
        if ((int)groups.size() >= slots) return false;
        const int32_t e = slot_end[next_slot()];
        return e < 0 || e <= k - 1;
    }
    // the next group (n batches) has been enqueued on the slot the pipeline returned
    void submitted(int slot, int32_t n)
    {
        groups.push_back({slot, next_first, n, false});
        slot_end[slot] = next_first + n;
        next_first += n;
        submit_count++;
    }
    // batch k is about to be handed over: groups that lie entirely before it are done
    void retire_before(int32_t k)
    {
        while (!groups.empty() && k >= groups.front().first + groups.front().n) groups.pop_front();
    }
};
