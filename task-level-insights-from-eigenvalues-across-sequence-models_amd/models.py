"""Drop-in ``models`` for the analysis path: the torch model families whose blocks produce the
inputs of the eigenvalue analysis (quirk Q1), runnable on MI355X without flash_attn / mamba_ssm / fla / Triton.

Same constructor configs, same attribute names and the same parameter names as the reference, so a ``.pth``
state_dict written by the reference's train.py (train.py:217-219) loads with ``load_state_dict`` unchanged, and
``analysis.eval_eig`` finds ``layer.attention.Wqkv``, ``layer.mamba.in_proj`` ... where it expects them:

    Transformer / TransformerBlock   models/transformer.py:22-161
    MHA, SelfAttention & co          models/attention.py:12-182
    MHNA, SelfNormAttention & co     models/norm_attention.py:17-259
    Mamba / MambaBlock / SSD(_LTI)   models/mamba.py:26-389
    MLP, GLU, LAMBDA, MATCH, ClassifierHead, TokenEmbeddings   models/common.py:15-178

Modules are created in the reference's order with the reference's initialisers, so that ``torch.manual_seed(seed)``
followed by construction draws the same parameters (the "init" spectra of eval_eig, quirk Q8).

Forward only (the analysis never back-propagates): the token mixers run in the HIP library through the C ABI
(``egs_linear_recurrence_fwd``, ``egs_softmax_attention_fwd``, ``egs_causal_conv1d_silu``, ``egs_layer_norm``); dense
layers, embeddings and dropout are torch ops on the GPU (rocBLAS plumbing).  There is no CPU path: a forward on CPU tensors
raises.
"""
import math

import torch
import torch.nn.functional as F
from torch import nn

from . import ops

__all__ = ["MATCH", "MLP", "GLU", "LAMBDA", "ClassifierHead", "TokenEmbeddings", "MHA", "MHNA", "SelfAttention",
           "SelfLinAttention", "SelfNormAttention", "TransformerBlock", "Transformer", "SSD", "SSD_LTI", "MambaBlock",
           "Mamba"]


# ---------------------------------------------------------------------------------------------------------------
# models/common.py
# ---------------------------------------------------------------------------------------------------------------
class MATCH(nn.Module):
    """LRA retrieval matching head (models/common.py:15-29)."""

    def __init__(self, input_dim, mlp_dim, output_dim):
        super().__init__()
        self.encoder = nn.Linear(input_dim, mlp_dim)
        self.middle = nn.Linear(mlp_dim, int(mlp_dim // 2))
        self.decoder = nn.Linear(int(mlp_dim // 2), output_dim)
        self.activation = nn.ReLU()

    def forward(self, x):
        return self.decoder(self.activation(self.middle(self.activation(self.encoder(x)))))


class MLP(nn.Module):
    """Transformer MLP (models/common.py:33-49): Linear, GELU, dropout, Linear, dropout."""

    def __init__(self, input_dim, mlp_dim, output_dim=None, dropout=0.0):
        super().__init__()
        self.output_dim = input_dim if output_dim is None else output_dim
        self.encoder = nn.Linear(input_dim, mlp_dim)
        self.decoder = nn.Linear(mlp_dim, self.output_dim)
        self.activation = nn.GELU()
        self.dropout = nn.Dropout(dropout)

    def forward(self, x):
        return self.dropout(self.decoder(self.dropout(self.activation(self.encoder(x)))))


class GLU(nn.Module):
    """models/common.py:51-59."""

    def __init__(self, input_dim, dropout=0.0):
        super().__init__()
        self.linear = nn.Linear(input_dim, input_dim * 2)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x):
        a, gate = self.linear(x).split(x.shape[2], dim=2)
        return self.dropout(a * torch.sigmoid(gate))


class LAMBDA(nn.Module):
    """Learned blend of a GLU and an MLP branch on one shared up-projection (models/common.py:61-86)."""

    def __init__(self, input_dim, init=0.5, dropout=0.0):
        super().__init__()
        self.dim = input_dim
        self.encoder = nn.Linear(input_dim, input_dim * 2)
        self.decoder = nn.Linear(input_dim * 2, input_dim)
        self.activation = nn.GELU()
        self.dropout = nn.Dropout(dropout)
        self.alpha = nn.Parameter(torch.ones((1,)) * (-math.log(1 / init - 1)))

    def glu(self, xz):
        x, z = torch.split(xz, [self.dim, self.dim], dim=-1)
        return x * torch.sigmoid(z)

    def mlp(self, x):
        return self.decoder(self.dropout(self.activation(x)))

    def forward(self, x):
        xz = self.encoder(x)
        a = torch.sigmoid(self.alpha)
        return self.dropout(a * self.glu(xz) + (1 - a) * self.mlp(xz))


class ClassifierHead(nn.Module):
    """Pooling + optional two-layer head (models/common.py:90-119)."""

    def __init__(self, input_dim, mlp_dim, num_classes, pooling):
        super().__init__()
        self.mlp_dim = mlp_dim
        self.pooling = pooling
        if self.mlp_dim != 0:
            self.encoder = nn.Linear(input_dim, mlp_dim)
            self.decoder = nn.Linear(mlp_dim, num_classes)
            self.activation = nn.ReLU()

    def forward(self, x):
        if self.pooling == "mean":
            x = torch.mean(x, dim=1)
        elif self.pooling == "max":
            x = torch.max(x, dim=1)[0]
        elif self.pooling == "sum":
            x = torch.sum(x, dim=1)
        elif self.pooling == "cls":
            x = x[:, 0, :]
        if self.mlp_dim != 0:
            x = self.decoder(self.activation(self.encoder(x)))
        return x


class TokenEmbeddings(nn.Module):
    """GPT-2 style token (+ optional position) embeddings (models/common.py:121-178)."""

    def __init__(self, embed_dim, vocab_size, max_position_embeddings, padding_idx=None, word_embed_proj_dim=None,
                 learnable=True, device="cuda", dtype="torch.float32"):
        super().__init__()
        self.device = device
        self.dtype = dtype
        if word_embed_proj_dim is None:
            self.word_embeddings = nn.Embedding(vocab_size, embed_dim, padding_idx=padding_idx)
            self.project_in = None
        else:
            self.word_embeddings = nn.Embedding(vocab_size, word_embed_proj_dim, padding_idx=padding_idx)
            self.project_in = nn.Linear(word_embed_proj_dim, embed_dim, bias=False)
        if not learnable:
            self.word_embeddings.weight.requires_grad = False
        self.max_position_embeddings = max_position_embeddings
        if self.max_position_embeddings > 0:
            self.position_embeddings = nn.Embedding(max_position_embeddings, embed_dim)

    def forward(self, input_ids, position_ids=None):
        seqlen = input_ids.shape[1]
        emb = self.word_embeddings(input_ids)
        if self.project_in is not None:
            emb = self.project_in(emb)
        if self.max_position_embeddings > 0:
            if position_ids is None:
                position_ids = torch.arange(seqlen, dtype=torch.long, device=input_ids.device)
            emb = emb + self.position_embeddings(position_ids)
        return emb


class LayerNorm(nn.LayerNorm):
    """nn.LayerNorm with the same parameters (weight, bias) whose forward is one HIP pass (egs_layer_norm)."""

    def forward(self, x):
        if len(self.normalized_shape) != 1 or self.normalized_shape[0] > 1024:
            return super().forward(x)
        return _hip(lambda a, w, b: ops.layer_norm(a, w, b, self.eps), x, self.weight, self.bias)


# ---------------------------------------------------------------------------------------------------------------
# token mixers: HIP through the C ABI
# ---------------------------------------------------------------------------------------------------------------
def _f32(t):
    return t if t.dtype == torch.float32 else t.float()


class _ForwardOnly(torch.autograd.Function):
    """The HIP mixers have no backward.  When autograd is recording and an input requires grad, the op is put on the
    tape with a backward that RAISES, so that ``loss.backward()`` through these models (they carry the names the
    reference's train.py imports) fails loudly instead of silently training only the layers behind the last mixer."""

    @staticmethod
    def forward(ctx, fn, *inputs):
        with torch.no_grad():
            return fn(*inputs)

    @staticmethod
    def backward(ctx, *grads):
        raise NotImplementedError("these models are forward-only (eigenvalue analysis): the HIP token mixers, LayerNorm "
                                  "and conv have no backward pass; train with the reference's models")


def _hip(fn, *tensors):
    """Run a HIP op on `tensors` outside autograd, keeping a raising backward node when gradients were requested."""
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors):
        return _ForwardOnly.apply(fn, *tensors)
    with torch.no_grad():
        return fn(*tensors)


class SelfLinAttention(nn.Module):
    """Causal linear attention with elu+1 feature maps (models/attention.py:58-83)."""

    def __init__(self, dropout=0.0):
        super().__init__()
        self.dropout = nn.Dropout(dropout)

    def forward(self, qk, v):
        b, l, h, dv = v.shape
        y = _hip(lambda qk_, v_: ops.linear_recurrence_fwd(_f32(qk_[:, :, 0]), _f32(qk_[:, :, 1]),
                                                             _f32(v_).reshape(b, l, h * dv), phi=True, normalise=True), qk, v)
        return self.dropout(y.view(b, l, h, dv))


class SelfAttention(nn.Module):
    """Causal softmax attention, scale 1/sqrt(d) (models/attention.py:12-35)."""

    def __init__(self, dropout=0.0):
        super().__init__()
        self.dropout = nn.Dropout(dropout)

    def forward(self, qk, v):
        b, l, h, dv = v.shape
        if self.training and self.dropout.p > 0.0:
            # dropout acts on the L x L attention matrix (:33): only this train-mode branch materialises it
            q, k = qk.unbind(dim=2)
            scores = torch.einsum("bthd,bshd->bhts", q, k * (1.0 / math.sqrt(q.shape[-1])))
            scores = scores + torch.triu(torch.full((l, l), -10000.0, device=scores.device), 1).to(scores.dtype)
            return torch.einsum("bhts,bshd->bthd", self.dropout(torch.softmax(scores, dim=-1, dtype=v.dtype)), v)
        y = _hip(lambda qk_, v_: ops.softmax_attention_fwd(_f32(qk_[:, :, 0]), _f32(qk_[:, :, 1]),
                                                             _f32(v_).reshape(b, l, h * dv)), qk, v)
        return y.view(b, l, h, dv)


class FlashAttention(SelfAttention):
    """The reference routes this through flash_attn in fp16 (models/attention.py:37-55); same function, fp32 here.
    Attention-matrix dropout is not part of the fused kernel: it is skipped in train mode as well."""

    def __init__(self, dropout=0.0):
        super().__init__(0.0)
        self.dropout_p = dropout

    def forward(self, qkv):                                            # (B, S, 3, H, D)
        b, l, _, h, d = qkv.shape
        return super().forward(qkv[:, :, :2], qkv[:, :, 2])


def init_offset(size, a=0.02, b=0.1, lo=8.0, hi=14.0):
    """Per-head offsets spread over [lo, hi] on an inverse-softplus grid (models/norm_attention.py:17-24)."""
    if size == 1:
        return torch.tensor([(hi - lo) / 2])
    x = torch.log(torch.expm1(torch.linspace(a, b, size)))
    x = (x - x.min()) / (x.max() - x.min())
    return x * abs(hi - lo) + lo


_NORM_FNS = {"exp": torch.exp, "elu": F.elu, "softplus": F.softplus, "sigmoid": torch.sigmoid}


class SelfNormAttention(nn.Module):
    """Linear attention with a learned per-token normaliser n_t = exp(-norm_fn(n + offset))
    (models/norm_attention.py:27-89; FlashNormAttention :91-156 is the same function through fla's chunked kernel)."""

    def __init__(self, norm_fn, approx_fn, num_heads=1, scale_B=False, offset=False, offset_init="uniform", dropout=0.0):
        super().__init__()
        if norm_fn not in _NORM_FNS:
            raise RuntimeError("normalization function {0} not implemented!".format(norm_fn))
        if approx_fn not in ("none", "elu"):
            raise RuntimeError("approximation function {0} not implemented!".format(approx_fn))
        self.norm_name, self.approx_name = norm_fn, approx_fn
        self.norm_fn = _NORM_FNS[norm_fn]
        if offset:
            if offset_init == "uniform":
                self.offset = nn.Parameter(init_offset(num_heads))
            elif offset_init == "exp":
                self.offset = nn.Parameter(torch.linspace(4, 9, num_heads))
            else:
                raise RuntimeError("Invalid init option {0}".format(offset_init))
        else:
            self.offset = None
        self.scale_B = scale_B
        self.dropout = nn.Dropout(dropout)

    def forward(self, qk, v, n):
        b, l, h, dv = v.shape
        scale = 1.0 / math.sqrt(qk.shape[-1]) if self.scale_B else 1.0
        def run(qk_, v_, n_, off_):
            raw = n_ + off_ if off_ is not None else n_
            w = torch.exp(-self.norm_fn(_f32(raw)))
            return ops.linear_recurrence_fwd(_f32(qk_[:, :, 0]), _f32(qk_[:, :, 1]), _f32(v_).reshape(b, l, h * dv), w=w,
                                             phi=(self.approx_name == "elu"), k_scale=scale)
        y = _hip(run, qk, v, n, self.offset)
        return self.dropout(y.view(b, l, h, dv))


FlashNormAttention = SelfNormAttention


def _depthwise_causal(conv, act, x, seqlen):
    """act(conv1d(x^T)[..., :L])^T with padding = kernel - 1 (models/attention.py:152-155): one fused HIP pass over the
    channel-last activations (``act`` is the module's nn.SiLU; kernels wider than 8 taps fall outside the ABI)."""
    if conv.kernel_size[0] > 8:
        raise NotImplementedError("depthwise conv kernels wider than 8 taps are not supported")
    return _hip(ops.causal_conv1d_silu, x, conv.weight, conv.bias)


class MHA(nn.Module):
    """Multi-head softmax / linear attention block (models/attention.py:85-182)."""

    def __init__(self, d_model, d_qk=None, num_heads=1, dim_conv=0, lin_att=True, dropout=0.0, bias=True, use_flash=True,
                 layer_idx=None, conv_type="full"):
        super().__init__()
        self.d_model = d_model
        self.conv_type = conv_type
        self.d_qk = d_model if d_qk is None else d_qk
        self.layer_idx = layer_idx
        self.num_heads = num_heads
        assert self.d_qk % num_heads == 0, "self.kdim must be divisible by num_heads"
        assert self.d_model % num_heads == 0, "self.vdim must be divisible by num_heads"
        self.head_dim = self.d_qk // num_heads
        self.v_dim = self.d_model // num_heads
        self.use_flash = use_flash
        self.Wqkv = nn.Linear(d_model, 2 * self.d_qk + d_model, bias=bias)
        if lin_att:
            self.inner_attn = SelfLinAttention(dropout)
        elif self.use_flash and self.head_dim == self.v_dim:
            self.inner_attn = FlashAttention(dropout)
        else:
            self.inner_attn = SelfAttention(dropout)
        self.out_proj = nn.Linear(d_model, d_model)
        self.use_conv = dim_conv > 0
        if dim_conv > 0:
            self.act = nn.SiLU()
            conv_dim = self.d_model + 2 * self.d_qk if conv_type == "full" else 2 * self.d_qk
            self.conv1d = nn.Conv1d(conv_dim, conv_dim, kernel_size=dim_conv, groups=conv_dim, padding=dim_conv - 1,
                                    bias=True)

    def forward(self, x):
        qkv = self.Wqkv(x)
        seqlen = qkv.shape[1]
        if self.use_conv:
            if self.conv_type == "full":
                qkv = _depthwise_causal(self.conv1d, self.act, qkv, seqlen)
            else:
                qk, v = torch.split(qkv, [2 * self.d_qk, self.d_model], dim=-1)
                qkv = torch.cat([_depthwise_causal(self.conv1d, self.act, qk, seqlen), v], dim=-1)
        if self.use_flash and self.head_dim == self.v_dim:
            # (three h d) packing; like the reference this calls inner_attn with ONE argument, which only the flash
            # softmax module accepts (a linear-attention MHA with use_flash and head_dim == v_dim fails there too)
            context = self.inner_attn(qkv.unflatten(-1, (3, self.num_heads, self.head_dim)))
        else:
            qk, v = torch.split(qkv, [2 * self.d_qk, self.d_model], dim=-1)
            context = self.inner_attn(qk.unflatten(-1, (2, self.num_heads, self.head_dim)),
                                      v.unflatten(-1, (self.num_heads, self.v_dim)))
        return self.out_proj(context.flatten(-2))


class MHNA(nn.Module):
    """Multi-head normalised linear attention block (models/norm_attention.py:159-259)."""

    def __init__(self, d_model, d_qk=None, num_heads=1, mode="attention", norm_fn="exp", approx_fn="none", scale_B=False,
                 offset=False, offset_init="uniform", learn_A=False, dim_conv=0, dropout=0.0, use_flash=False, bias=True,
                 layer_idx=None, conv_type="full"):
        super().__init__()
        self.d_model = d_model
        self.conv_type = conv_type
        self.d_qk = d_model if d_qk is None else d_qk
        self.layer_idx = layer_idx
        self.num_heads = num_heads
        assert self.d_qk % num_heads == 0, "self.kdim must be divisible by num_heads"
        assert self.d_model % num_heads == 0, "self.vdim must be divisible by num_heads"
        self.head_dim = self.d_qk // num_heads
        self.v_dim = self.d_model // num_heads
        self.Wvqkn = nn.Linear(d_model, d_model + 2 * self.d_qk + num_heads, bias=bias)
        if mode != "attention":
            raise RuntimeError("Invalid mode option {0}".format(mode))
        self.inner_attn = SelfNormAttention(norm_fn, approx_fn, num_heads, scale_B, offset, offset_init, dropout)
        self.out_proj = nn.Linear(d_model, d_model)
        self.use_conv = dim_conv > 0
        if dim_conv > 0:
            self.act = nn.SiLU()
            conv_dim = self.d_model + 2 * self.d_qk if conv_type == "full" else 2 * self.d_qk
            self.conv1d = nn.Conv1d(conv_dim, conv_dim, kernel_size=dim_conv, groups=conv_dim, padding=dim_conv - 1,
                                    bias=True)

    def forward(self, x):
        seqlen = x.shape[1]
        vqk, n = torch.split(self.Wvqkn(x), [self.d_model + 2 * self.d_qk, self.num_heads], dim=-1)
        if self.use_conv:
            if self.conv_type == "full":
                vqk = _depthwise_causal(self.conv1d, self.act, vqk, seqlen)
            else:
                v, qk = torch.split(vqk, [self.d_model, 2 * self.d_qk], dim=-1)
                vqk = torch.cat([v, _depthwise_causal(self.conv1d, self.act, qk, seqlen)], dim=-1)
        v, qk = torch.split(vqk, [self.d_model, 2 * self.d_qk], dim=-1)
        context = self.inner_attn(qk.unflatten(-1, (2, self.num_heads, self.head_dim)),
                                  v.unflatten(-1, (self.num_heads, self.v_dim)), n)
        return self.out_proj(context.flatten(-2))


# ---------------------------------------------------------------------------------------------------------------
# models/transformer.py
# ---------------------------------------------------------------------------------------------------------------
class TransformerBlock(nn.Module):
    """norm -> attention -> dropout -> +skip;  norm (the SAME LayerNorm) -> mixer -> gate / residual
    (models/transformer.py:22-111)."""

    def __init__(self, hidden_dim, cfg, layer_idx=None):
        super().__init__()
        d_model, d_qk, num_heads = hidden_dim, cfg["state_dim"], cfg["num_heads"]
        att_dropout, mixer, mixer_dim = cfg["att_dropout"], cfg["mixer"], cfg["mixer_dim"]
        dropout, norm = cfg["dropout"], cfg["norm"]
        use_flash = cfg["use_flash"] if "use_flash" in cfg else False
        use_gate = cfg["use_gate"] if "use_gate" in cfg else False
        conv_type = cfg["conv_type"] if "conv_type" in cfg else "full"
        self.attention_fn = cfg["attention_fn"]
        if self.attention_fn in ["sm-attention", "lin-attention"]:
            dim_conv = cfg["dim_conv"] if "dim_conv" in cfg else 0
            self.attention = MHA(d_model, d_qk, num_heads, dim_conv=dim_conv,
                                 lin_att=self.attention_fn == "lin-attention", dropout=att_dropout, use_flash=use_flash,
                                 conv_type=conv_type)
        elif self.attention_fn in ["norm-attention"]:
            self.attention = MHNA(d_model, d_qk, num_heads, cfg["mode"], cfg["norm_fn"], cfg["approx_fn"], cfg["scale_B"],
                                  cfg["offset"], cfg["offset_init"], cfg["learn_A"], cfg["dim_conv"], dropout=att_dropout,
                                  use_flash=use_flash, layer_idx=layer_idx, conv_type=conv_type)
        self.use_gate = use_gate
        if self.use_gate:
            self.Wz = nn.Linear(d_model, d_model)
            nn.init.constant_(self.Wz.bias, 1.0)
            nn.init.xavier_uniform_(self.Wz.weight, gain=0.1)
        if mixer == "mlp":
            self.mixer, self.drop_skip = MLP(hidden_dim, mixer_dim, dropout=dropout), False
        elif mixer == "glu":
            self.mixer, self.drop_skip = GLU(hidden_dim), False
        elif mixer == "hybrid":
            self.mixer, self.drop_skip = LAMBDA(hidden_dim, init=0.2, dropout=dropout), False
        elif mixer == "none":
            self.mixer, self.drop_skip = nn.Identity(hidden_dim), True
        else:
            raise RuntimeError("{0} mixer not implemented yet!".format(mixer))
        if norm != "layer":
            raise RuntimeError("{0} norm not implemented yet!".format(norm))
        self.norm = LayerNorm(hidden_dim)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x):
        z = self.Wz(x) if self.use_gate else None
        x = self.dropout(self.attention(self.norm(x))) + x
        y = self.mixer(self.norm(x))
        if self.drop_skip:
            return y * F.silu(z) if self.use_gate else y
        return (x + y) * F.silu(z) if self.use_gate else x + y


class Transformer(nn.Module):
    """models/transformer.py:113-161."""

    def __init__(self, cfg):
        super().__init__()
        hidden_dim, num_layers = cfg["hidden_dim"], cfg["num_layers"]
        self.dual = cfg["dual"]
        self.classify = cfg["classifier"]
        if cfg["embedding"]:
            self.encoder = TokenEmbeddings(hidden_dim, cfg["vocab_size"], cfg["max_pos_embed"])
        else:
            self.encoder = nn.Linear(cfg["input_dim"], hidden_dim)
        self.layers = nn.Sequential(*[TransformerBlock(hidden_dim, cfg, idx) for idx in range(num_layers)])
        if self.classify:
            self.classifier = ClassifierHead(hidden_dim, cfg["mixer_dim"], cfg["output_dim"], cfg["pooling"])
        else:
            self.decoder = nn.Linear(hidden_dim, cfg["output_dim"], bias=False)
        if self.dual:
            self.match = MATCH(cfg["output_dim"] * 2, cfg["mixer_dim"], cfg["output_dim"])
        if cfg["norm"] != "layer":
            raise RuntimeError("{0} norm not implemented yet!".format(cfg["norm"]))
        self.norm = LayerNorm(hidden_dim)
        self.dropout = nn.Dropout(cfg["dropout"])

    def forward(self, x):
        x = self.norm(self.layers(self.dropout(self.encoder(x))))
        if self.classify:
            x = self.classifier(x)
            if self.dual:
                x1, x2 = torch.split(x, int(x.shape[0] / 2))
                x = self.match(torch.concatenate((x1, x2), dim=1))
        else:
            x = self.decoder(x)
        return x


# ---------------------------------------------------------------------------------------------------------------
# models/mamba.py
# ---------------------------------------------------------------------------------------------------------------
class _SSDBase(nn.Module):
    """Parameters shared by SSD and SSD_LTI, created in the reference's order (models/mamba.py:26-109, :152-250)."""

    def _build(self, d_model, d_state, d_conv, expand, headdim, ngroups, A_init_range, dt_min, dt_max, dt_init_floor,
               dt_limit, learnable_init_states, bias, chunk_size, layer_idx, device, dtype, dt_cols, lti):
        fk = {"device": device, "dtype": dtype}
        self.d_model, self.d_state, self.expand = d_model, d_state, expand
        self.d_inner = self.expand * self.d_model
        self.headdim, self.ngroups = headdim, ngroups
        assert self.d_inner % self.headdim == 0
        self.nheads = self.d_inner // self.headdim
        self.dt_limit = dt_limit
        self.learnable_init_states = learnable_init_states
        self.chunk_size = chunk_size
        self.layer_idx = layer_idx
        if lti:
            assert (self.d_state * self.ngroups) % self.nheads == 0
            self.khead_dim = (self.d_state * self.ngroups) // self.nheads
        n_dt = self.nheads if dt_cols == "nheads" else self.ngroups
        self.in_proj = nn.Linear(self.d_model, int(self.d_inner + 2 * self.ngroups * self.d_state + n_dt), bias=bias, **fk)
        if self.learnable_init_states:
            self.init_states = nn.Parameter(torch.zeros(self.nheads, self.headdim, self.d_state, **fk))
        dt = torch.exp(torch.rand(self.nheads, **fk) * (math.log(dt_max) - math.log(dt_min)) + math.log(dt_min))
        dt = torch.clamp(dt, min=dt_init_floor)
        self.dt_bias = nn.Parameter(dt + torch.log(-torch.expm1(-dt)))      # inverse softplus
        if not lti:
            assert A_init_range[0] > 0 and A_init_range[1] >= A_init_range[0]
            A = torch.empty(self.nheads, dtype=torch.float32, device=device).uniform_(*A_init_range)
            self.A_log = nn.Parameter(torch.log(A).to(dtype=dtype))
        self.D = nn.Parameter(torch.ones(self.nheads, device=device))
        if lti:
            A = torch.empty((self.nheads,), **fk).uniform_(-8, -2)          # eigenvalues in (0.88, 0.999)
            self.A = nn.Parameter(A)
            self.register_buffer("beta", torch.ones((self.nheads,), **fk))
        if d_conv > 0:
            self.use_conv = True
            self.act = nn.SiLU()
            conv_dim = self.d_inner + 2 * self.ngroups * self.d_state
            self.conv1d = nn.Conv1d(conv_dim, conv_dim, kernel_size=d_conv, groups=conv_dim, padding=d_conv - 1, bias=True)
        else:
            self.use_conv = False
        self.out_proj = nn.Linear(self.d_inner, self.d_model, bias=bias, **fk)

    def _split(self, u, n_dt):
        seqlen = u.shape[1]
        xBC, dt = torch.split(self.in_proj(u), [self.d_inner + 2 * self.ngroups * self.d_state, n_dt], dim=-1)
        dt = F.softplus(dt + self.dt_bias)
        if self.use_conv:
            xBC = _depthwise_causal(self.conv1d, self.act, xBC, seqlen)
        gn = self.ngroups * self.d_state
        x, Bm, Cm = torch.split(xBC, [self.d_inner, gn, gn], dim=-1)
        return x, Bm, Cm, dt

    def _scan(self, x, decay, vscale, Bm, Cm):
        """y_t = C_t^T S_t + D x_t,  S_t = decay_t S_{t-1} + B_t (vscale_t x_t)^T  (the SSD recurrence that
        mamba_ssm's mamba_chunk_scan_combined evaluates chunk-wise; z = None, no dt_softplus, default dt_limit)."""
        if self.learnable_init_states:
            raise NotImplementedError("learnable_init_states is not supported by the HIP scan")
        b, l, _ = x.shape

        def run(x_, decay_, vscale_, b_, c_, d_):
            return ops.linear_recurrence_fwd(_f32(c_).unflatten(-1, (self.ngroups, self.d_state)),
                                             _f32(b_).unflatten(-1, (self.ngroups, self.d_state)), _f32(x_),
                                             decay=decay_, vscale=vscale_, skip=_f32(d_).repeat_interleave(self.headdim),
                                             cols_per_decay=self.headdim)
        return _hip(run, x, decay, vscale, Bm, Cm, self.D)

    def _clamp_dt(self, dt):
        """mamba_chunk_scan_combined clamps dt to dt_limit BEFORE it forms exp(dt A) and dt x: decay and input scale both
        see the clamped value."""
        if self.dt_limit != (0.0, float("inf")):
            return dt.clamp(self.dt_limit[0], self.dt_limit[1])
        return dt


class SSD(_SSDBase):
    """Mamba-2 SSD mixer without the gate branch (models/mamba.py:26-150)."""

    def __init__(self, d_model, d_state=64, d_conv=4, expand=1, headdim=32, ngroups=1, A_init_range=(1, 16), dt_min=0.001,
                 dt_max=0.1, dt_init_floor=1e-4, dt_limit=(0.0, float("inf")), learnable_init_states=False, bias=False,
                 chunk_size=256, layer_idx=None, device=None, dtype=None):
        super().__init__()
        self._build(d_model, d_state, d_conv, expand, headdim, ngroups, A_init_range, dt_min, dt_max, dt_init_floor,
                    dt_limit, learnable_init_states, bias, chunk_size, layer_idx, device, dtype, "nheads", False)

    def forward(self, u, seq_idx=None):
        x, Bm, Cm, dt = self._split(u, self.nheads)
        A = -torch.exp(self.A_log)
        dt = self._clamp_dt(_f32(dt))
        decay = torch.exp(dt * _f32(A))                              # the eigenvalue lambda_t of eval_eig.py:176-190
        return self.out_proj(self._scan(x, decay, dt, Bm, Cm))


class SSD_LTI(_SSDBase):
    """Pseudo-LTI variant: input-independent decay exp(beta * -softplus(A)), dt folded into B
    (models/mamba.py:152-291)."""

    def __init__(self, d_model, d_state=64, d_conv=4, expand=1, headdim=32, ngroups=1, A_init_range=(1, 16), dt_min=0.001,
                 dt_max=0.1, dt_init_floor=1e-4, dt_limit=(0.0, float("inf")), learnable_init_states=False, bias=False,
                 chunk_size=256, layer_idx=None, device=None, dtype=None):
        super().__init__()
        self._build(d_model, d_state, d_conv, expand, headdim, ngroups, A_init_range, dt_min, dt_max, dt_init_floor,
                    dt_limit, learnable_init_states, bias, chunk_size, layer_idx, device, dtype, "ngroups", True)

    def forward(self, u, seq_idx=None):
        batch, seqlen, _ = u.shape
        x, Bm, Cm, dt = self._split(u, self.ngroups)                 # dt: (B, L, nheads) after the dt_bias broadcast
        beta = self.beta.expand(batch, seqlen, -1)
        A = -F.softplus(self.A)
        Bm = dt.repeat_interleave(self.khead_dim, dim=-1) * Bm       # (B, L, d_state)
        beta = self._clamp_dt(_f32(beta)).contiguous()
        decay = torch.exp(beta * _f32(A)).contiguous()
        # the reference reshapes B, C as "b l (n g) -> b l n g" with n = ngroups (:283-284): ngroups leading
        return self.out_proj(self._scan(x, decay, beta, Bm, Cm))


class MambaBlock(nn.Module):
    """(pre)norm -> mamba -> GELU -> dropout -> [GLU] -> dropout -> +skip -> (post)norm (models/mamba.py:293-336)."""

    def __init__(self, version, hidden_dim, state_dim, headdim, conv_dim, expansion, dropout, glu, norm, prenorm,
                 pseudoLTI):
        super().__init__()
        if version == "mamba2":
            cls = SSD_LTI if pseudoLTI else SSD
            self.mamba = cls(d_model=hidden_dim, d_state=state_dim, d_conv=conv_dim, expand=expansion, headdim=headdim)
        elif version == "mamba1":
            raise RuntimeError("Mamba-1 (selective_scan) is outside the analysed path: eval_eig only supports mamba2")
        else:
            raise RuntimeError("Non supported version")
        self.glu = GLU(hidden_dim) if glu else None
        self.activation = nn.GELU()
        self.dropout = nn.Dropout(dropout)
        if norm == "layer":
            self.norm = LayerNorm(hidden_dim)
        elif norm == "batch":
            raise RuntimeError("dimensions don't agree for batch norm to work")
        self.prenorm = prenorm

    def forward(self, x):
        skip = x
        if self.prenorm:
            x = self.norm(x)
        x = self.dropout(self.activation(self.mamba(x)))
        if self.glu is not None:
            x = self.glu(x)
        x = self.dropout(x) + skip
        if not self.prenorm:
            x = self.norm(x)
        return x


class Mamba(nn.Module):
    """models/mamba.py:338-389."""

    def __init__(self, cfg):
        super().__init__()
        hidden_dim = cfg["hidden_dim"]
        headdim = hidden_dim // cfg["num_heads"]
        pseudo_lti = cfg["pseudoLTI"] if "pseudoLTI" in cfg else False
        if cfg["token_embedding"]:
            self.encoder = TokenEmbeddings(hidden_dim, cfg["vocab_size"], 0)
        else:
            self.encoder = nn.Linear(cfg["input_dim"], hidden_dim)
        self.blocks = nn.Sequential(*[
            MambaBlock(cfg["version"], hidden_dim, cfg["state_dim"], headdim, cfg["conv_dim"], cfg["expansion"],
                       cfg["dropout"], cfg["glu"], cfg["norm"], cfg["prenorm"], pseudo_lti)
            for _ in range(cfg["num_layers"])])
        self.decoder = nn.Linear(hidden_dim, cfg["output_dim"])
        self.pooling = cfg["pooling"]
        self.dual = cfg["dual"]
        if self.dual:
            self.match = MATCH(cfg["output_dim"] * 2, cfg["output_dim"])   # as in the reference (:373): 2 arguments

    def forward(self, x):
        x = self.blocks(self.encoder(x))
        if self.pooling == "mean":
            x = torch.mean(x, dim=1)
        elif self.pooling == "max":
            x = torch.max(x, dim=1)[0]
        elif self.pooling == "last":
            x = x[:, -1, :]
        x = self.decoder(x)
        if self.dual:
            x1, x2 = torch.split(x, int(x.shape[0] / 2))
            x = self.match(torch.concatenate((x1, x2), dim=1))
        return x
