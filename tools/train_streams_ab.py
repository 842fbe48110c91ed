"""CLIP ViT-B/16 training step (B = 256): both towers on one stream vs the text tower (forward and backward) on a side stream.  Same-process
alternating A/B.   python tools/train_streams_ab.py"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def main():
    from multimodal_amd.schedule import set_schedule
    from multimodal_amd.models.clip import clip_vit_b16
    from multimodal_amd.modules.losses.contrastive_loss_with_temperature import ContrastiveLossWithTemperature
    from multimodal_amd.utils.synthetic import clip_batch

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = clip_vit_b16().to(dev).train()
    loss_fn = ContrastiveLossWithTemperature().to(dev)
    opt = torch.optim.SGD(list(model.parameters()) + list(loss_fn.parameters()), lr=1e-4)
    images, ids = clip_batch(256)
    images, ids = images.to(dev), ids.to(dev)

    def step():
        opt.zero_grad(set_to_none=True)
        o = model(images, ids)
        loss = loss_fn(o.embeddings_a, o.embeddings_b)
        loss.backward()
        opt.step()

    # full-size check first: the same three steps from the same state in both modes -> the same losses and gradients
    import copy

    state = (copy.deepcopy(model.state_dict()), copy.deepcopy(loss_fn.state_dict()))
    traj = {}
    for flag in (False, True):
        model.load_state_dict(state[0])
        loss_fn.load_state_dict(state[1])
        set_schedule(train_side_stream=flag)
        vals = []
        for _ in range(3):
            opt.zero_grad(set_to_none=True)
            o = model(images, ids)
            loss = loss_fn(o.embeddings_a, o.embeddings_b)
            loss.backward()
            torch.cuda.synchronize()
            vals.append((float(loss), float(model.encoder_a.conv.weight.grad.double().sum()), float(model.encoder_b.projection.weight.grad.double().sum()),
                         float(model.encoder_b.encoder.layers[0].linear1.weight.grad.double().abs().sum())))
            opt.step()
        traj[flag] = vals
    print("one stream :", traj[False])
    print("side stream:", traj[True])
    print("first step identical:", traj[False][0] == traj[True][0], flush=True)
    for _ in range(3):
        step()
    for rnd in range(3):
        for flag in (False, True):
            set_schedule(train_side_stream=flag)
            step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(4):
                step()
            torch.cuda.synchronize()
            print("text tower on a side stream " if flag else "one stream                  ", round((time.perf_counter() - t0) / 4 * 1e3, 2), "ms", flush=True)
    set_schedule(train_side_stream=True)


if __name__ == "__main__":
    main()
