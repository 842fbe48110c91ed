"""Generates tests/golden/zero_shot.npz by calling the REFERENCE'S OWN functions: `_zero_shot_classifier`, `_accuracy`
and `run_imagenet_zero_shot` of /root/reference/examples/flava/native/utils.py:100-160 and `compute_recall` of
/root/reference/examples/flava/coco_zero_shot.py:24-31, on seeded inputs, with a table-lookup stand-in for the towers (what is pinned
here are the read-outs, not the encoders).  The two example files import hydra, omegaconf, torchvision datasets and the example
package `flava.data` at module scope; those are stubbed exactly like torchvision / iopath in _ref_shim.py (none is called by the four
functions).  Build container only:

    python -m tests.golden.make_golden_zero_shot
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.golden import _ref_shim  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
REF = _ref_shim.REFERENCE_ROOT

CLASSNAMES = ["tench", "goldfish", "great white shark", "tiger shark", "hammerhead", "electric ray", "stingray", "cock", "hen", "ostrich",
              "brambling"]
TEMPLATES = [lambda c: f"a photo of a {c}.", lambda c: f"a bad photo of a {c}.", lambda c: f"a sculpture of a {c}.",
             lambda c: f"a photo of the hard to see {c}.", lambda c: f"a low resolution photo of the {c}."]
VOCAB, L, E = 211, 48, 96


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def load_reference_examples():
    _ref_shim.install()
    _mod("flava")
    _mod("flava.data")
    _mod("flava.data.imagenet_zeroshot_data", imagenet_classnames=CLASSNAMES, openai_imagenet_template=TEMPLATES)
    _mod("flava.data.transforms", default_image_pretraining_transforms=None, default_text_transform=None)
    _mod("hydra")
    _mod("hydra.utils", instantiate=lambda *a, **k: None)
    _mod("omegaconf", DictConfig=dict, OmegaConf=types.SimpleNamespace(from_cli=None, load=None, merge=None))
    tv = sys.modules.get("torchvision") or _mod("torchvision")
    ds = _mod("torchvision.datasets", CocoCaptions=object)
    tv.datasets = ds
    out = []
    for rel, name in (("examples/flava/native/utils.py", "_ref_flava_native_utils"), ("examples/flava/coco_zero_shot.py", "_ref_coco_zero_shot")):
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        out.append(mod)
    return out


def text_transform(texts):
    """Stand-in for the HF tokenizer of the example: deterministic ids from the characters, {'input_ids': LongTensor [T, L]}."""
    rows = []
    for t in texts:
        ids = [(7 * ord(ch) + 13 * i) % VOCAB for i, ch in enumerate(t)][:L]
        rows.append(ids + [0] * (L - len(ids)))
    return {"input_ids": torch.tensor(rows, dtype=torch.long)}


class TableModel:
    """model(x, action=...) of the example trainer: text -> sum of a seeded table's rows, image -> the given features."""

    def __init__(self, g):
        self.table = torch.randn(VOCAB, E, generator=g)

    def __call__(self, x, action):
        if action == "encode_text":
            return self.table[x].sum(1)  # a fresh tensor: the reference normalises it in place
        return x.clone()


def main():
    utils, coco = load_reference_examples()
    g = torch.Generator().manual_seed(0)
    model = TableModel(g)
    out = {}
    C, T = len(CLASSNAMES), len(TEMPLATES)
    dev = torch.device("cpu")
    classifier = utils._zero_shot_classifier(model, dev, text_transform)  # REFERENCE: [E, C]
    ids = torch.stack([text_transform([t(c) for t in TEMPLATES])["input_ids"] for c in CLASSNAMES])  # [C, T, L]
    prompts = torch.stack([model(ids[c], "encode_text") for c in range(C)])  # [C, T, E] raw prompt embeddings
    # 8 batches of image features; the reference's loop stops after 6 (utils.py:149-150)
    NB, n = 8, 16
    labels = torch.randint(0, C, (NB, n), generator=g)
    feats = torch.randn(NB, n, E, generator=g) * 3 + classifier.t()[labels] * 4
    loader = [{"image": feats[i], "label": labels[i]} for i in range(NB)]
    results = utils.run_imagenet_zero_shot(model, loader, dev, text_transform)  # REFERENCE
    flat = feats.reshape(-1, E)
    f = flat / flat.norm(dim=-1, keepdim=True)
    logits = 100.0 * f @ classifier
    target = labels.reshape(-1).clone()
    target[: target.numel() // 2] = logits[: target.numel() // 2].argmax(1)  # a mix of hits and misses
    acc = utils._accuracy(logits, target, topk=(1, 5, 10))  # REFERENCE
    out.update(prompts=prompts.numpy(), prompt_ids=ids.numpy(), table=model.table.numpy(), classifier=classifier.numpy(),
               feats=flat.numpy(), logits=logits.numpy(), target=target.numpy(), acc=np.asarray(acc), batch_feats=feats.numpy(),
               batch_labels=labels.numpy(), run_top1=np.float64(results["imagenet-zeroshot-val-top1"]),
               run_top5=np.float64(results["imagenet-zeroshot-val-top5"]))
    # retrieval: compute_recall of the COCO example on the example's own normalised similarity (coco_zero_shot.py:84-95)
    M = 61
    img = torch.randn(M, E, generator=g)
    txt = img + 5.0 * torch.randn(M, E, generator=g)
    a = torch.nn.functional.normalize(img, dim=-1)
    b = torch.nn.functional.normalize(txt, dim=-1)
    sim = a @ b.t()
    rec = [float(coco.compute_recall(s, k=k)) for s in (sim, sim.t()) for k in (1, 5)]  # REFERENCE
    out.update(img=img.numpy(), txt=txt.numpy(), sim=sim.numpy(), recall=np.asarray(rec))
    np.savez_compressed(os.path.join(HERE, "zero_shot.npz"), **out)
    print("acc", acc, "run", results, "recall", rec)


if __name__ == "__main__":
    main()
