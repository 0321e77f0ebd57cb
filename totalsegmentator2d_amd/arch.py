"""Architecture descriptor for the 2-D nnU-Net ``PlainConvUNet`` hot path.

The reference never builds the network itself: ``nnUNetPredictor.initialize_from_trained_model_folder``
(reference call site ``ts2d/core/inference/nnu.py:164-165``) reads ``plans.json`` and instantiates
``dynamic_network_architectures.architectures.unet.PlainConvUNet`` from
``plans['configurations'][cfg]['architecture']['arch_kwargs']`` (third-party, nnunetv2ml==2.6.2,
``pyproject.toml:25``).  This module is the host-side mirror of that
descriptor: it validates the subset the MI355X engine implements, expands it into the flat layer
*program* the C-ABI engine executes, enumerates the parameter tensors in the exact order the engine's
weight blob uses, and does the FLOP/byte accounting BASELINE.md quotes.
"""
from __future__ import annotations

from dataclasses import dataclass, asdict
from typing import Dict, List, Sequence, Tuple

OP_CONV3X3 = 0    # Conv2d 3x3 pad 1 (+bias) -> InstanceNorm2d(affine) -> LeakyReLU   (K1/K2/K3/K6)
OP_CONVT2X2 = 1   # ConvTranspose2d k=2 s=2 (+bias), no norm / activation             (K5)
OP_HEAD1X1 = 2    # seg_layers[-1]: Conv2d 1x1 (+bias) -> logits                       (K7)


@dataclass
class UNetArch:
    """The ``arch_kwargs`` of a ``PlainConvUNet`` (2-D), plus input channels and number of heads.

    Field names follow the upstream keyword names so that a real ``plans.json`` maps 1:1
    (see :func:`UNetArch.from_plans`)."""
    input_channels: int = 2
    num_classes: int = 18
    n_stages: int = 8
    features_per_stage: Sequence[int] = (32, 64, 128, 256, 512, 512, 512, 512)
    kernel_sizes: Sequence[Sequence[int]] = ((3, 3),) * 8
    strides: Sequence[Sequence[int]] = ((1, 1),) + ((2, 2),) * 7
    n_conv_per_stage: Sequence[int] = (2,) * 8
    n_conv_per_stage_decoder: Sequence[int] = (2,) * 7
    conv_bias: bool = True
    norm_eps: float = 1e-5
    norm_affine: bool = True
    leaky_slope: float = 0.01

    # ------------------------------------------------------------------ constructors
    @staticmethod
    def canonical(input_channels: int = 2, num_classes: int = 18, n_stages: int = 8,
                  base: int = 32, max_features: int = 512) -> "UNetArch":
        """The nnU-Net 2-D planner output SURVEY.md section 8 assumes (8 stages for 512x512, 9 for 1024x1024)."""
        feats = [min(base * 2 ** i, max_features) for i in range(n_stages)]
        return UNetArch(input_channels=input_channels, num_classes=num_classes, n_stages=n_stages,
                        features_per_stage=tuple(feats), kernel_sizes=((3, 3),) * n_stages,
                        strides=((1, 1),) + ((2, 2),) * (n_stages - 1),
                        n_conv_per_stage=(2,) * n_stages, n_conv_per_stage_decoder=(2,) * (n_stages - 1))

    @staticmethod
    def from_plans(plans: dict, configuration: str, input_channels: int, num_classes: int) -> "UNetArch":
        """Map ``plans.json`` -> descriptor.  Upstream layout (nnU-Net v2.3+):
        ``plans['configurations'][cfg]['architecture'] = {network_class_name, arch_kwargs, _kw_requires_import}``."""
        cfg = plans['configurations'][configuration]
        while 'inherits_from' in cfg and 'architecture' not in cfg:
            cfg = plans['configurations'][cfg['inherits_from']]
        arch = cfg['architecture']
        cls = arch['network_class_name']
        if not cls.endswith('PlainConvUNet'):
            raise NotImplementedError(f"network class '{cls}' is not supported by the MI355X engine (PlainConvUNet only)")
        kw = arch['arch_kwargs']
        if not str(kw['conv_op']).endswith('Conv2d'):
            raise NotImplementedError(f"conv_op {kw['conv_op']} is not supported (2-D only)")
        if not str(kw['norm_op']).endswith('InstanceNorm2d'):
            raise NotImplementedError(f"norm_op {kw['norm_op']} is not supported")
        if kw.get('dropout_op') is not None:
            pass  # dropout is the identity at inference
        if not str(kw['nonlin']).endswith('LeakyReLU'):
            raise NotImplementedError(f"nonlin {kw['nonlin']} is not supported")
        nk = kw.get('norm_op_kwargs') or {}
        n_stages = int(kw['n_stages'])

        def _per_stage(v, n):
            return tuple(v) if isinstance(v, (list, tuple)) else (v,) * n
        return UNetArch(
            input_channels=input_channels, num_classes=num_classes, n_stages=n_stages,
            features_per_stage=tuple(int(f) for f in _per_stage(kw['features_per_stage'], n_stages)),
            kernel_sizes=tuple(tuple(k) if isinstance(k, (list, tuple)) else (k, k) for k in kw['kernel_sizes']),
            strides=tuple(tuple(s) if isinstance(s, (list, tuple)) else (s, s) for s in kw['strides']),
            n_conv_per_stage=tuple(int(n) for n in _per_stage(kw['n_conv_per_stage'], n_stages)),
            n_conv_per_stage_decoder=tuple(int(n) for n in _per_stage(kw['n_conv_per_stage_decoder'], n_stages - 1)),
            conv_bias=bool(kw.get('conv_bias', True)),
            norm_eps=float(nk.get('eps', 1e-5)), norm_affine=bool(nk.get('affine', True)),
            leaky_slope=float((kw.get('nonlin_kwargs') or {}).get('negative_slope', 0.01)))

    # ------------------------------------------------------------------ validation
    def validate(self) -> None:
        n = self.n_stages
        if n < 2:
            raise ValueError("n_stages must be >= 2")
        for name, v, ln in (('features_per_stage', self.features_per_stage, n), ('kernel_sizes', self.kernel_sizes, n),
                            ('strides', self.strides, n), ('n_conv_per_stage', self.n_conv_per_stage, n),
                            ('n_conv_per_stage_decoder', self.n_conv_per_stage_decoder, n - 1)):
            if len(v) != ln:
                raise ValueError(f"{name} must have {ln} entries, found {len(v)}")
        for k in self.kernel_sizes:
            if tuple(k) != (3, 3):
                raise NotImplementedError(f"kernel size {tuple(k)} is not supported (3x3 only)")
        if tuple(self.strides[0]) != (1, 1):
            raise NotImplementedError("first stage must have stride 1")
        for s in self.strides[1:]:
            # nnU-Net's planner pools every axis separately (reference: the plan is whatever the model folder holds,
            # ts2d/core/inference/nnu.py:164-165): (2, 2) until one axis is exhausted, then (2, 1) / (1, 2)
            if len(s) != 2 or any(int(v) not in (1, 2) for v in s):
                raise NotImplementedError(f"stride {tuple(s)} is not supported (1 or 2 per axis)")
        if not self.conv_bias or not self.norm_affine:
            raise NotImplementedError("conv_bias=False / affine=False are not supported")
        if self.input_channels < 1 or self.num_classes < 1:
            raise ValueError("input_channels and num_classes must be positive")
        # (any width: the engine rounds a stage up to a multiple of 32 with zero weights in the added channels - exact, csrc/engine.hip pad_arch;
        #  the head kernel reads at most 64 channels)
        if any(f < 1 for f in self.features_per_stage):
            raise ValueError("features_per_stage must be positive")
        if self.features_per_stage[0] > 64:
            raise NotImplementedError(f"features_per_stage[0] = {self.features_per_stage[0]}: the 1x1 head kernel supports at most 64 channels")
        if any(c < 1 for c in list(self.n_conv_per_stage) + list(self.n_conv_per_stage_decoder)):
            raise ValueError("n_conv_per_stage* must be >= 1")

    def level_shifts(self) -> List[Tuple[int, int]]:
        """(log2 of the cumulative stride along H, along W) of every stage: a level-s tensor of an H x W input has
        extent (H >> ly, W >> lx)."""
        out, ly, lx = [], 0, 0
        for s in self.strides:
            ly += int(s[0]) // 2
            lx += int(s[1]) // 2
            out.append((ly, lx))
        return out

    def extent(self, level: int, H: int, W: int) -> Tuple[int, int]:
        ly, lx = self.level_shifts()[level]
        return H >> ly, W >> lx

    @property
    def divisors(self) -> Tuple[int, int]:
        """H and W must be multiples of these (products of the strides per axis)."""
        ly, lx = self.level_shifts()[-1]
        return 2 ** ly, 2 ** lx

    @property
    def divisor(self) -> int:
        """The larger of :attr:`divisors` (both, for an isotropic plan)."""
        return max(self.divisors)

    # ------------------------------------------------------------------ program
    def program(self) -> List[dict]:
        """Flat op list in execution order.  Each entry: op, name, cin (and cin_skip for the virtual concat), cout,
        stride (sy, sx), level (0 = full resolution), src / skip tensor names, dst name.  ``torch.cat((up, skip), 1)``
        (upsampled first - SURVEY K6) is represented by ``cin`` (up) + ``cin_skip`` and never materialised."""
        self.validate()
        ops: List[dict] = []
        cur, cin = 'input', self.input_channels
        for s in range(self.n_stages):
            f = self.features_per_stage[s]
            for i in range(self.n_conv_per_stage[s]):
                stride = tuple(int(v) for v in self.strides[s]) if (i == 0 and s > 0) else (1, 1)
                dst = f'enc{s}.c{i}'
                ops.append(dict(op=OP_CONV3X3, name=dst, src=cur, skip=None, cin=cin, cin_skip=0, cout=f,
                                stride=stride, level=s, dst=dst,
                                key=f'encoder.stages.{s}.0.convs.{i}'))
                cur, cin = dst, f
        skips = {s: f'enc{s}.c{self.n_conv_per_stage[s] - 1}' for s in range(self.n_stages)}
        for j in range(self.n_stages - 1):          # decoder stage j handles skip level n-2-j
            lvl = self.n_stages - 2 - j
            f = self.features_per_stage[lvl]
            up = f'dec{lvl}.up'
            # transposed conv: kernel = stride = the stride of the encoder stage below (upstream UNetDecoder)
            ops.append(dict(op=OP_CONVT2X2, name=up, src=cur, skip=None, cin=cin, cin_skip=0, cout=f,
                            stride=tuple(int(v) for v in self.strides[lvl + 1]),
                            level=lvl, dst=up, key=f'decoder.transpconvs.{j}'))
            cur = up
            for i in range(self.n_conv_per_stage_decoder[j]):
                dst = f'dec{lvl}.c{i}'
                ops.append(dict(op=OP_CONV3X3, name=dst, src=cur, skip=skips[lvl] if i == 0 else None,
                                cin=f, cin_skip=f if i == 0 else 0, cout=f, stride=(1, 1), level=lvl, dst=dst,
                                key=f'decoder.stages.{j}.convs.{i}'))
                cur = dst
            cin = f
        ops.append(dict(op=OP_HEAD1X1, name='head', src=cur, skip=None, cin=cin, cin_skip=0, cout=self.num_classes,
                        stride=(1, 1), level=0, dst='logits', key=f'decoder.seg_layers.{self.n_stages - 2}'))
        return ops

    def param_specs(self) -> List[Tuple[str, Tuple[int, ...]]]:
        """(state-dict key, shape) of every tensor of the weight blob, in blob order (PyTorch layouts:
        conv ``[Cout,Cin,3,3]``, convT ``[Cin,Cout,sy,sx]`` (kernel = stride), head ``[K,Cin,1,1]``)."""
        specs: List[Tuple[str, Tuple[int, ...]]] = []
        for op in self.program():
            k, cin, cout = op['key'], op['cin'] + op['cin_skip'], op['cout']
            if op['op'] == OP_CONV3X3:
                specs += [(f'{k}.conv.weight', (cout, cin, 3, 3)), (f'{k}.conv.bias', (cout,)),
                          (f'{k}.norm.weight', (cout,)), (f'{k}.norm.bias', (cout,))]
            elif op['op'] == OP_CONVT2X2:
                specs += [(f'{k}.weight', (cin, cout) + tuple(op['stride'])), (f'{k}.bias', (cout,))]
            else:
                specs += [(f'{k}.weight', (cout, cin, 1, 1)), (f'{k}.bias', (cout,))]
        return specs

    def n_params(self) -> int:
        n = 0
        for _, shp in self.param_specs():
            c = 1
            for d in shp:
                c *= d
            n += c
        return n

    # ------------------------------------------------------------------ accounting
    def work(self, H: int, W: int, act_bytes: int = 4) -> Dict[str, float]:
        """Algorithmic FLOPs (2*MAC; bias/norm/activation not counted) and layer-wise activation bytes
        (one read of every op's input, one write of its output) for ONE slice of H x W."""
        macs = 0
        act = 0
        per_layer = []
        for op in self.program():
            h, w = self.extent(op['level'], H, W)
            cin, cout = op['cin'] + op['cin_skip'], op['cout']
            sy, sx = op['stride']
            if op['op'] == OP_CONV3X3:
                m = h * w * cout * cin * 9
                hin, win = h * sy, w * sx
                rd, wr = hin * win * cin * act_bytes, h * w * cout * act_bytes
            elif op['op'] == OP_CONVT2X2:
                m = h * w * cout * cin          # each output pixel = one tap
                rd, wr = (h // sy) * (w // sx) * cin * act_bytes, h * w * cout * act_bytes
            else:
                m = h * w * cout * cin
                rd, wr = h * w * cin * act_bytes, h * w * cout * 4
            macs += m
            act += rd + wr
            per_layer.append(dict(name=op['name'], macs=m, rd=rd, wr=wr))
        return dict(flops=2.0 * macs, macs=float(macs), act_bytes=float(act),
                    io_bytes=float(H * W * (self.input_channels * 4 + self.num_classes * 4)),
                    weight_bytes=float(self.n_params() * 4), per_layer=per_layer)

    def to_dict(self) -> dict:
        d = asdict(self)
        for k, v in d.items():
            if isinstance(v, tuple):
                d[k] = [list(x) if isinstance(x, tuple) else x for x in v]
        return d
