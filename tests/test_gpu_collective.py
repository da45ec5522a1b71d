"""-m gpu: the counter all-reduce of the multi-GPU path through the C ABI (RCCL).  One GPU on the
test box: a one-rank communicator, whose sum must be the identity; the two-rank arithmetic of the
same reduction is covered on CPU by tests/test_sharding_gloo.py."""
import numpy as np
import pytest

from thermite_amd import capi, refdata, synth

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(300)
def test_single_rank_rccl_allreduce_is_identity(data_dir):
    t = refdata.load_reference(data_dir + "/GRCh38-2020-A-chrM.fasta", data_dir + "/GRCh38-2020-A-chrM.gtf")
    bases, off, _ = synth.simulate_reads(t, 3000, 91, sub_rate=0.02, indel_rate=0.004, stream=21)
    a = capi.Aligner(capi.Index(t), capi.CI_OPTS)
    a.align_batch(bases, off)
    before = a.counters()
    assert before[0] == 3000 and before[3] > 0
    comm = capi.Comm(capi.comm_unique_id(), 1, 0, 0)
    a.counters_allreduce(comm)
    assert np.array_equal(a.counters(), before)
    # a second batch keeps accumulating on top of the reduced totals
    a.align_batch(bases, off)
    a.counters_allreduce(comm)
    assert np.array_equal(a.counters(), 2 * before)
    comm.close()
    a.close()


def test_allreduce_argument_checks(data_dir):
    """null handles are THM_ERR_INVALID_ARG, never a crash; a communicator that was freed is not touched again"""
    import ctypes as C

    L = capi.lib()
    t = refdata.load_reference(data_dir + "/test_ref.fasta", data_dir + "/test_ref.gtf")
    a = capi.Aligner(capi.Index(t), capi.CI_OPTS)
    comm = capi.Comm(capi.comm_unique_id(), 1, 0, 0)
    assert L.thm_counters_allreduce(None, comm.h) == capi.ERR_INVALID_ARG
    assert L.thm_counters_allreduce(a.h, None) == capi.ERR_INVALID_ARG
    assert L.thm_comm_unique_id(None) == capi.ERR_INVALID_ARG
    out = C.c_void_p()
    assert L.thm_comm_create(None, 1, 0, 0, C.byref(out)) == capi.ERR_INVALID_ARG and not out.value
    assert L.thm_comm_create(capi._ptr(capi.comm_unique_id()), 1, 0, 0, None) == capi.ERR_INVALID_ARG
    L.thm_comm_free(None)  # a no-op
    a.counters_allreduce(comm)  # still usable after the failed calls
    comm.close()
    a.close()


def test_comm_argument_checks():
    uid = np.zeros(128, np.uint8)
    for nranks, rank in ((0, 0), (2, 2), (1, -1)):
        with pytest.raises(capi.ThermiteError) as e:
            capi.Comm(uid, nranks, rank, 0)
        assert e.value.code == capi.ERR_INVALID_ARG
    with pytest.raises(capi.ThermiteError) as e:
        capi.Comm(uid, 1, 0, 99)
    assert e.value.code == capi.ERR_NO_DEVICE
