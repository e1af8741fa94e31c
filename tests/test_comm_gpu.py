"""GPU: the RCCL binding of the C ABI (hpn_comm_*, hpn_allreduce_u64) on a 1-rank communicator.
The 2+-rank path is the same call on every rank; its reduction logic is covered on CPU by
tests/test_shard_gloo.py and on hardware by the driver's multi-GPU bench."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_single_rank_allreduce_is_identity_and_keeps_the_tally():
    import torch
    import highperformancengs_amd as hp
    from highperformancengs_amd import _lib
    import orc
    assert torch.cuda.is_available()
    ctx = hp.Context(0)
    uid = hp.comm_unique_id()
    assert len(uid) == _lib.UNIQUE_ID_BYTES
    ctx.comm_init(0, 1, uid)
    seq, qual, off = orc.synth_soa(11, 0, 5000, 40, 150)
    dq = torch.from_numpy(qual).cuda()
    do = torch.from_numpy(off.astype(np.int64)).cuda()
    ctx.fastq_tally_dev(dq, do, 5000)
    ctx.allreduce_u64(ctx.tally_devptr(), _lib.W_BAD + 1)  # in place on the context's stream
    got = ctx.fastq_tally_fetch()
    rc, want = orc.count_soa(qual, off)
    s = want.summary()
    assert np.array_equal(got.seqlen, want.seqlen) and (got.total, got.q20, got.q30) == (s.sum, s.q20, s.q30)
    v = torch.arange(1000, dtype=torch.int64, device="cuda")
    ctx.allreduce_u64(v, 1000)
    ctx.sync()
    assert torch.equal(v, torch.arange(1000, dtype=torch.int64, device="cuda"))
    with pytest.raises(hp.HpnError):
        ctx.comm_init(0, 1, uid)  # already initialised
    ctx.close()


def test_allreduce_without_comm_is_a_state_error():
    import torch
    import highperformancengs_amd as hp
    from highperformancengs_amd import _lib
    ctx = hp.Context(0)
    v = torch.zeros(4, dtype=torch.int64, device="cuda")
    with pytest.raises(hp.HpnError) as e:
        ctx.allreduce_u64(v, 4)
    assert e.value.status == _lib.E_STATE
    ctx.close()


def test_grouped_allreduce_of_one_process():
    """hpn_comm_init_all / hpn_allreduce_u64_all: the C tools' collective (one process, one communicator per context).
    On the one-GPU box: a group of one context is the identity and keeps the tally; two contexts on ONE device are
    refused (one RCCL rank per device) without making anything -- the tools then add the vectors on the host."""
    import torch
    import highperformancengs_amd as hp
    from highperformancengs_amd import _lib, api
    import orc
    a, b = hp.Context(0), hp.Context(0)
    with pytest.raises(hp.HpnError) as e:
        api.comm_init_all([a, b])
    assert e.value.status == _lib.E_ARG
    with pytest.raises(hp.HpnError) as e:
        api.allreduce_u64_all([a], [a.tally_devptr()], 8)     # nothing was made by the refused call
    assert e.value.status == _lib.E_STATE
    api.comm_init_all([a])
    lib = api.comm_library()
    assert "rccl" in lib
    # inside a torch process the binding must have found the RCCL torch already mapped, not a second one beside it
    maps = open("/proc/self/maps").read()
    assert len({l.split()[-1] for l in maps.splitlines() if "librccl" in l}) == 1, lib
    seq, qual, off = orc.synth_soa(12, 0, 4000, 30, 151)
    dq, do = torch.from_numpy(qual).cuda(), torch.from_numpy(off.astype(np.int64)).cuda()
    a.fastq_tally_dev(dq, do, 4000)
    api.allreduce_u64_all([a], [a.tally_devptr()], _lib.TALLY_WORDS)
    got = a.fastq_tally_fetch()
    rc, want = orc.count_soa(qual, off)
    s = want.summary()
    assert np.array_equal(got.seqlen, want.seqlen) and (got.total, got.q20, got.q30) == (s.sum, s.q20, s.q30)
    a.close()
    b.close()
