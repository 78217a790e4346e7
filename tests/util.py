"""Shared helpers for the parity tests (test infrastructure)."""
import ctypes as C

import numpy as np
import torch


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def max_abs(a, b) -> float:
    return float((a.double().cpu() - b.double().cpu()).abs().max())


def iou(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.cpu().bool(), b.cpu().bool()
    u = (a | b).sum().item()
    return 1.0 if u == 0 else (a & b).sum().item() / u


from sam_pt_amd.synth import disc_queries, synthetic_clip  # noqa: E402,F401


def make_fake_device_predictor(sd, cfg, max_decode_batch=3):
    """A CPU predictor with the device-path API of sam_pt_amd.SamPredictor (encode_frames / track_decode incl. ragged
    batches), computed by the oracle — lets the fused host logic run without a GPU."""
    import torch
    from oracle import sam_ref as R

    class FakeDevicePredictor(R.SamPredictorRef):
        def __init__(self):
            super().__init__(sd, cfg)
            self.model.max_decode_batch = max_decode_batch
            self.calls = []

        def encode_frames(self, frames, chw=True, batch_events=None):
            x = R.preprocess(cfg, frames.float())
            return torch.cat([R.image_encoder(sd, cfg, x[i:i + 1]) for i in range(len(x))])

        def track_decode(self, feats, pts, labels, k, n_pos_first, refine_iters, iou_thr, size_hw, out_logits, out_score,
                         k_item=None, npos_item=None):
            self.calls.append((feats.shape[0], k, n_pos_first, k_item is not None))
            for i in range(feats.shape[0]):
                ki = int(k_item[i]) if k_item is not None else k
                pi = int(npos_item[i]) if npos_item is not None else n_pos_first
                self.features, self.original_size, self.input_size = feats[i:i + 1], tuple(size_hw), tuple(size_hw)
                pc, pl = pts[i:i + 1, :ki], labels[i:i + 1, :ki]
                kw = dict(multimask_output=False, return_logits=True)
                low = None
                if n_pos_first >= 0:
                    _, _, low = self.predict_torch(pc[:, :pi], pl[:, :pi], None, None, **kw)
                ml, iou, low = self.predict_torch(pc, pl, None, low, **kw)
                for _ in range(refine_iters):
                    msk = ml[0, 0] > 0
                    if msk.sum() < 2:
                        break
                    yx = msk.nonzero()
                    box = torch.tensor([[yx[:, 1].min(), yx[:, 0].min(), yx[:, 1].max(), yx[:, 0].max()]], dtype=torch.float)
                    ml, iou, low = self.predict_torch(pc, pl, box, low, **kw)
                out_score[i] = iou[0, 0]
                out_logits[i] = ml[0, 0] if float(iou[0, 0]) >= iou_thr else -float("inf")

    return FakeDevicePredictor()
