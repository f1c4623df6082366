"""Inputs of the survey-F8 full-step fixtures (tools/gen_golden.py:step_case), regenerated from the same seeds; shared
by the CPU test (oracle vs fixture) and the GPU test (HIP trainer vs fixture)."""
from oracle.weights import det_tensor, fill_state_dict, rect_masks


def full_step_setup(tag, nb, hw):
    """Weights / inputs / noise stream of tools/gen_golden.py:step_case, regenerated from the same seeds."""
    from graphecho_amd.models.fpnseg import FPN, Discriminator
    from graphecho_amd.models.graph_matching import GModule

    fpn_sd = fill_state_dict(FPN([2, 4, 23, 3], 4, 3, back_bone="resnet").state_dict(), seed=1)
    gm_sd = fill_state_dict(GModule(256, 4, "cpu").state_dict(), seed=6)
    dis_sd = {name: fill_state_dict(Discriminator(grad_reverse_lambda=0.02).state_dict(), seed=20 + i)
              for i, name in enumerate(("p2", "p3", "p4", "p5"))}
    xs = det_tensor(f"step.{tag}.xs", (nb, 3, hw, hw), "uniform")
    xt = det_tensor(f"step.{tag}.xt", (nb, 3, hw, hw), "uniform")
    masks = rect_masks(nb, 4, hw, hw, seed=3)
    draws = []

    def noise_fn(n, d):
        draws.append((n, d))
        return det_tensor(f"noise.{len(draws) - 1}", (n, d))

    return fpn_sd, gm_sd, dis_sd, xs, xt, masks, noise_fn, draws


def temporal_step_setup(tag="c5", nb=2, hw=256, t=16):
    """Weights / inputs / noise stream of tools/gen_golden.py:temporal_case (BASELINE config 5 as train_cardiac_uda.py
    runs it: FPN(in_channel=1, back_bone="VGG16"), one source + one target clip of t frames), from the same seeds."""
    from graphecho_amd.models.fpnseg import FPN, Discriminator
    from graphecho_amd.models.graph_matching import GModule
    from graphecho_amd.models.TGCN import TGCN

    sds = {"Net": fill_state_dict(FPN([2, 4, 23, 3], 4, 1, back_bone="VGG16").state_dict(), seed=1),
           "Graph": fill_state_dict(GModule(256, 4, "cpu").state_dict(), seed=6),
           "tgcn_p5": fill_state_dict(TGCN(256, 256, (t, hw // 32, hw // 32), 10, 10,
                                           transport_method="sinkhorn_distance").state_dict(), seed=7)}
    for i, name in enumerate(("p2", "p3", "p4", "p5")):
        sds["Dis_P" + name[1]] = fill_state_dict(Discriminator(grad_reverse_lambda=0.02).state_dict(), seed=20 + i)
    xs = det_tensor(f"temporal.{tag}.xs", (nb, 1, hw, hw), "uniform")
    xt = det_tensor(f"temporal.{tag}.xt", (nb, 1, hw, hw), "uniform")
    masks = rect_masks(nb, 4, hw, hw, seed=3)
    cm = rect_masks(t, 4, hw, hw, seed=5).permute(1, 2, 3, 0).unsqueeze(0).contiguous()
    cm[..., 1::4] = 0
    clips = {"source": det_tensor(f"temporal.{tag}.cs", (1, 1, hw, hw, t), "uniform"),
             "target": det_tensor(f"temporal.{tag}.ct", (1, 1, hw, hw, t), "uniform"), "masks": cm}
    draws = []

    def noise_fn(n, d):
        draws.append((n, d))
        return det_tensor(f"noise.{len(draws) - 1}", (n, d))

    return sds, xs, xt, masks, clips, noise_fn, draws
