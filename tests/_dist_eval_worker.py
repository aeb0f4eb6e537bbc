"""Worker of tests/test_gpu_dist_eval_eig.py (not a test): one rank of a sharded ``eval_eig`` call on a toy model.
Started as a plain subprocess with RANK / WORLD_SIZE / MASTER_* in the environment, exactly like a process that
``torchrun launch.py --config ... --analysis_config ...`` would start: eval_eig joins the process group by itself."""
import copy
import importlib
import json
import os
import sys

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
B, L, D, DQK, H, NL, VOCAB = 10, 24, 16, 8, 2, 2, 32      # 10 sequences over 2 ranks: 5 + 5; over 4: 3 + 3 + 3 + 1


class _Attn(nn.Module):
    def __init__(self):
        super().__init__()
        self.head_dim = DQK // H
        self.Wqkv = nn.Linear(D, 2 * DQK + D)


class _Block(nn.Module):
    def __init__(self, p):
        super().__init__()
        self.attention = _Attn()
        self.mix = nn.Linear(D, D)
        self.drop = nn.Dropout(p)

    def forward(self, x):
        return x + self.drop(torch.tanh(self.mix(x)))


class ToyTransformer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.encoder = nn.Embedding(VOCAB, D)
        self.layers = nn.ModuleList([_Block(cfg.get("dropout", 0.0)) for _ in range(cfg["num_layers"])])


def main():
    spec = json.loads(sys.argv[1])
    os.chdir(spec["cwd"])
    pkg = importlib.import_module("task-level-insights-from-eigenvalues-across-sequence-models_amd")
    args = {"model": {"layer": "transformer", "num_layers": NL, "num_heads": H, "hidden_dim": D, "state_dim": DQK,
                      "seq_len": L, "attention_fn": spec["attention_fn"], "dropout": spec["dropout"]},
            "train": {"lr": 0.01}, "dataset": {"name": "TOY"}, "seed": 1919}
    rng = np.random.default_rng(3)
    X = torch.from_numpy(rng.integers(0, VOCAB, (B, L)))
    conf = {"batch_size": B, "save_path": spec["cwd"] + "/"}
    if spec.get("exchange"):
        conf["exchange"] = spec["exchange"]
    out = pkg.eval_eig(copy.deepcopy(args), conf, None, None, [(X, torch.zeros(B), None)], spec["ckpt"], 0.5,
                       model_classes={"transformer": ToyTransformer})
    names = ["eig", "eig_init", "pct", "pct_init", "pct_phase", "pct_phase_init"]
    np.savez(os.path.join(spec["cwd"], "out.npz"), **dict(zip(names, out)))
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
