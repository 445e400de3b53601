"""Hardware queues of the contexts' streams (mld_contexts_concurrent; the search in mld_create): the HIP runtime multiplexes
the streams of a process on a few hardware queues, and two contexts on one queue lose their whole overlap.  A new context
therefore takes a stream that runs side by side with the stream of every live context of its device."""
import pytest
import torch

from mono_lidar_depth_amd import capi

from helpers import make_estimator

pytestmark = pytest.mark.gpu


def test_contexts_get_hardware_queues_of_their_own():
    P = capi.params_c0()
    # streams the host application (here: torch's stream pool) holds: they take hardware queues of the same pool
    pool = [torch.cuda.Stream(device="cuda:0") for _ in range(5)]
    for s in pool:
        with torch.cuda.stream(s):
            torch.zeros(16, device="cuda:0").add_(1)
    torch.cuda.synchronize()
    a = make_estimator(P)
    b = make_estimator(P)
    c = make_estimator(P)
    # two (the bench's schedule) and three contexts are pairwise concurrent, whatever was created before them
    assert a.concurrentWith(b) and b.concurrentWith(a)
    assert a.concurrentWith(c) and b.concurrentWith(c)
    assert not a.concurrentWith(a)
    # contexts come and go (the secondary bench legs create and close a dozen): later pairs are still concurrent
    # (three live contexts at a time here: the runtime's default is four hardware queues per process and priority)
    b.close()
    c.close()
    for _ in range(4):
        x = make_estimator(P)
        y = make_estimator(P)
        x.close()
        z = make_estimator(P)
        assert y.concurrentWith(z) and a.concurrentWith(z) and a.concurrentWith(y)
        y.close()
        z.close()
    a.close()
    del pool
