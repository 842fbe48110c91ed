

def test_two_tower_schedule_policy():
    """The grouped (layer-locked, one stream) schedule is chosen only where it measured faster: towers of equal depth, a dominant image tower,
    every projection pair of a layer ONE persistent launch — ViT-B/16 at the benchmark batch; B/32, L/14 and small batches keep two streams."""
    from multimodal_amd.models.clip import clip_vit_b16, clip_vit_b32, clip_vit_l14
    from multimodal_amd.models.clip._transformer import two_stacks_groupable

    def ok(model, S, B):
        return two_stacks_groupable(model.encoder_a.encoder, B * S, model.encoder_b.encoder, B * 77)

    b16, b32, l14 = clip_vit_b16(), clip_vit_b32(), clip_vit_l14()
    assert ok(b16, 197, 256) and ok(b16, 197, 512)
    assert not ok(b32, 50, 256) and not ok(b32, 50, 512)      # comparable towers: two streams overlap better
    assert not ok(l14, 257, 256)                                  # 24 + 12 layers: not layer-locked to the end
    assert not ok(b16, 197, 128) and not ok(b16, 197, 8)          # too few tiles for grouped launches
