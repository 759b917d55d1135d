"""Host-side logic of legion_amd.engine that needs no GPU."""
from legion_amd import engine


class _FakePipeline(engine.Pipeline):
    def __init__(self, group_size):
        self.group_size = group_size
        self.calls = []

    def submit(self, counter0, mode=engine.TRAINMODE, n_active=None):
        self.calls.append((counter0, n_active))
        return len(self.calls) % 2


def test_run_range_groups_and_tail():
    p = _FakePipeline(4)
    last = p.run_range(3, 10)
    assert p.calls == [(3, 4), (7, 4), (11, 2)]
    assert last[1:] == (11, 2)


def test_run_range_wraps_over_the_seed_set_without_straddling():
    """Another epoch over the same seeds (the reference's schedule: GetLocalBatchId, ipc_service.cu:213-228): batch indices
    are taken modulo the epoch and a launch group never straddles the wrap."""
    p = _FakePipeline(4)
    p.run_range(6, 12, wrap=8)                  # batches 6,7 | 0..3 | 4..7 | 0,1
    assert p.calls == [(6, 2), (0, 4), (4, 4), (0, 2)]
    p = _FakePipeline(4)
    p.run_range(0, 16, wrap=8)                  # a multiple of the group size: only full groups
    assert p.calls == [(0, 4), (4, 4), (0, 4), (4, 4)]
    covered = []
    p = _FakePipeline(5)
    p.run_range(13, 37, wrap=20)
    for c0, n in p.calls:
        assert 0 <= c0 and c0 + n <= 20 and 1 <= n <= 5
        covered += list(range(c0, c0 + n))
    assert covered == [(13 + k) % 20 for k in range(37)]
