"""CLIPSeg (+) UNet logit ensemble: fused prediction and the validation alpha grid search
(predict_CLIPseg.py:501-525, eval_CLIPseg.py:656-723; best_alpha.txt holds the reference's result, 10.0)."""
import numpy as np
import torch

from ._lib import lib, ptr, require_gpu, stream


def fuse_predict(clip_logits, unet_logits, alpha, return_fused=False):
    """clip_logits [N,C,hc,wc] (e.g. 352x352), unet_logits [N,C,H,W] -> argmax mask int64 [N,H,W] (and fused logits)."""
    require_gpu()
    c, u = clip_logits.contiguous().float(), unet_logits.contiguous().float()
    N, C, hc, wc = c.shape
    _, _, H, W = u.shape
    pred = torch.empty((N, H, W), dtype=torch.int64, device=u.device)
    fused = torch.empty((N, C, H, W), dtype=torch.float32, device=u.device) if return_fused else None
    lib().call("egm_ensemble_fuse", ptr(c), ptr(u), float(alpha), N, C, hc, wc, H, W, ptr(pred), ptr(fused), stream())
    return (pred, fused) if return_fused else pred


def search_best_alpha(clip_logits_list, unet_logits_list, labels_list, search_scale=(0.1, 10.0), search_step=100, num_classes=2):
    """-> (best_alpha, best_miou, miou per alpha).  Global confusion matrix over all images per alpha; first maximum wins,
    like the reference's strict `>` update."""
    require_gpu()
    alphas = np.linspace(search_scale[0], search_scale[1], search_step)
    dev = unet_logits_list[0].device
    a_dev = torch.tensor(alphas, dtype=torch.float32, device=dev)
    hist = torch.zeros(search_step * num_classes * num_classes, dtype=torch.int64, device=dev)
    for c, u, t in zip(clip_logits_list, unet_logits_list, labels_list):
        c, u = c.contiguous().float().to(dev), u.contiguous().float().to(dev)
        t = torch.as_tensor(t).to(dev).to(torch.int64).reshape(u.shape[0], u.shape[2], u.shape[3]).contiguous()
        lib().call("egm_ensemble_alpha_hist", ptr(c), ptr(u), ptr(t), ptr(a_dev), search_step, u.shape[0], u.shape[1], c.shape[2], c.shape[3],
                   u.shape[2], u.shape[3], ptr(hist), stream())
    miou = torch.empty(search_step, dtype=torch.float32, device=dev)
    lib().call("egm_ensemble_miou", ptr(hist), search_step, num_classes, ptr(miou), stream())
    m = miou.cpu().numpy()
    best, best_miou = 0.0, 0.0
    for a, v in zip(alphas, m):
        if v > best_miou:
            best_miou, best = float(v), float(a)
    return best, best_miou, m
