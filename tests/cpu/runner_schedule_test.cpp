// Simulates a trainer against RunnerSchedule (legion_amd/csrc/runner_schedule.h): the server loop of GPURunner::RunOnce with the
// `views` hand-over, every group-size pattern, 2-4 slots, and a trainer that releases as LATE as the protocol allows.  Invariant:
// a lane is never overwritten while the trainer may still read the batch that lives in it.
//   g++ -O1 -std=c++17 runner_schedule_test.cpp -o t && ./t
#include <cstdio>
#include <cstdlib>
#include <random>

#include "../../legion_amd/csrc/runner_schedule.h"

static int run(int slots, int lanes, int total, unsigned seed, bool ragged)
{
    std::mt19937 rng(seed);
    RunnerSchedule s;
    s.reset(slots);
    // what lives where: owner[slot][lane] = batch id whose data the lane holds (-1 none)
    std::vector<std::vector<int>> owner(slots, std::vector<int>(lanes, -1));
    int released = 0;            // batches 0 .. released-1 have been released (synchronize() called)
    int posted = 0;              // batches 0 .. posted-1 have been posted to the trainer
    auto plan = [&](int first) { int n = ragged ? 1 + (int)(rng() % lanes) : lanes; return std::min(n, total - first); };
    auto submit = [&](int k) -> int {       // returns 0 ok, 1 violation
        const int slot = s.next_slot();
        const int n = plan(s.next_first);
        for (int l = 0; l < lanes; l++) {   // the sampler overwrites EVERY lane of the slot
            const int b = owner[slot][l];
            if (b >= 0 && b >= released) { printf("VIOLATION: slot %d lane %d holds unreleased batch %d (released %d, k %d)\n", slot, l, b, released, k); return 1; }
            owner[slot][l] = -1;
        }
        for (int l = 0; l < n; l++) owner[slot][l] = s.next_first + l;
        s.submitted(slot, n);
        return 0;
    };
    if (submit(-1)) return 1;                                   // PrepareServing: the first group, before any trainer exists
    for (int k = 0; k < total; k++) {
        // the server waits for the token of batch k: tokens available = 2 + released.  The adversarial trainer releases only when
        // the server would otherwise block for ever (it must hold at most the batches posted and not released)
        while (k >= 2 + released) {
            if (released >= posted) { printf("DEADLOCK at k %d\n", k); return 1; }
            released++;                                         // synchronize() of the oldest batch it holds
        }
        // sometimes the trainer is quick instead: releases everything it has
        if (rng() % 4 == 0) released = posted;
        s.retire_before(k);
        while (s.next_first < total && s.may_submit(k))
            if (submit(k)) return 1;
        if (s.groups.empty() || k < s.groups.front().first || k >= s.groups.front().first + s.groups.front().n) { printf("LOST TRACK at k %d\n", k); return 1; }
        const RunnerSchedule::Group& g = s.groups.front();
        if (owner[g.slot][k - g.first] != k) { printf("WRONG LANE: batch %d not in slot %d lane %d\n", k, g.slot, k - g.first); return 1; }
        posted = k + 1;                                         // handed over as a view of that lane
    }
    return 0;
}

int main()
{
    int bad = 0, runs = 0;
    for (int slots = 2; slots <= 4; slots++)
        for (int lanes : {1, 2, 3, 5, 8, 64})
            for (int ragged = 0; ragged < 2; ragged++)
                for (unsigned seed = 1; seed <= 40; seed++) {
                    bad += run(slots, lanes, 500 + (int)(seed * 7), seed * 977 + slots * 31 + lanes, ragged != 0);
                    runs++;
                }
    printf("%d simulations, %d failed\n", runs, bad);
    return bad ? 1 : 0;
}
