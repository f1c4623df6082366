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
