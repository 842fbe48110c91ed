"""The phased half-batch schedule (schedule.phases = 2; CLIP._towers_phased, mmamd_stream_set_cus): two half-batches on two streams with half the
chip's CUs each.  Same kernels, same per-sample arithmetic -> the embeddings must equal the one-stream grouped schedule's BIT FOR BIT, for even
and odd batch sizes and every phase lead; the CU budgets must be gone afterwards; a HIP-graph capture of the phased step replays the same values."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def clip_b16():
    from multimodal_amd.models.clip import clip_vit_b16

    torch.manual_seed(0)
    return clip_vit_b16().to("cuda").eval()


def _batch(B):
    from multimodal_amd.utils.synthetic import clip_batch

    images, ids = clip_batch(B)
    return images.to("cuda").to(torch.bfloat16), ids.to("cuda")


@pytest.mark.parametrize("B,lead", [(256, 4), (255, 2), (254, 9), (256, 40)])
def test_phased_equals_grouped_bitwise(clip_b16, B, lead):
    from multimodal_amd import ops
    from multimodal_amd.schedule import get_schedule, set_schedule

    images, ids = _batch(B)
    prev = get_schedule()
    try:
        with torch.no_grad():
            set_schedule(two_tower="grouped", phases=1)
            ref = clip_b16(images, ids)
            ra, rb = ref.embeddings_a.clone(), ref.embeddings_b.clone()
            set_schedule(phases=2, phase_lead=lead)
            assert clip_b16._phased(clip_b16.encoder_a, images, ids)
            for _ in range(2):  # second call: packed copies warm, allocator blocks reused across the two streams
                out = clip_b16(images, ids)
                torch.cuda.synchronize()
                assert torch.equal(out.embeddings_a, ra) and torch.equal(out.embeddings_b, rb)
        assert ops.stream_cus(torch.cuda.current_stream()) == 256  # budgets cleared
    finally:
        set_schedule(two_tower=prev.two_tower, phases=prev.phases, phase_lead=prev.phase_lead)


def test_phased_under_graph_capture(clip_b16):
    from multimodal_amd.schedule import get_schedule, set_schedule

    images, ids = _batch(256)
    prev = get_schedule()
    try:
        with torch.no_grad():
            set_schedule(two_tower="grouped", phases=2)
            ref = clip_b16(images, ids)
            ra, rb = ref.embeddings_a.clone(), ref.embeddings_b.clone()
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                clip_b16(images, ids)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=s):
                    out = clip_b16(images, ids)
                g.replay()
                torch.cuda.synchronize()
            assert torch.equal(out.embeddings_a, ra) and torch.equal(out.embeddings_b, rb)
    finally:
        set_schedule(two_tower=prev.two_tower, phases=prev.phases, phase_lead=prev.phase_lead)


def test_small_batches_keep_the_one_stream_schedule(clip_b16):
    from multimodal_amd.schedule import get_schedule, set_schedule

    images, ids = _batch(8)
    prev = get_schedule()
    try:
        set_schedule(phases=2)
        assert not clip_b16._phased(clip_b16.encoder_a, images, ids)  # halves too small for grouped persistent launches on 128 CUs
        with torch.no_grad():
            out = clip_b16(images, ids)
        assert out.embeddings_a.shape == (8, 512)
    finally:
        set_schedule(phases=prev.phases)


def test_stream_cu_budget_api():
    from multimodal_amd import ops

    s = torch.cuda.Stream()
    assert ops.stream_cus(s) == 256
    ops.stream_set_cus(s, 128)
    assert ops.stream_cus(s) == 128
    with pytest.raises(ops.MmamdError):
        ops.stream_set_cus(s, 100)  # whole XCD slices only
    ops.stream_set_cus(s, 0)
    assert ops.stream_cus(s) == 256
    # a GEMM on a budgeted stream gives the same result as on the whole chip
    torch.manual_seed(1)
    a = torch.randn(4096, 768, device="cuda").to(torch.bfloat16)
    w = torch.randn(2304, 768, device="cuda").to(torch.bfloat16)
    ref = ops.gemm_bf16(a, w)
    ops.stream_set_cus(s, 64)
    try:
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            got = ops.gemm_bf16(a, w)
        s.synchronize()
    finally:
        ops.stream_set_cus(s, 0)
    assert torch.equal(got, ref)
