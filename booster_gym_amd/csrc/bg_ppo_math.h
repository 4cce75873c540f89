// Per-sample PPO actor loss, forward and analytic backward (reference utils/runner.py:145-174 + utils/utils.py:47-52), shared by
// bg_ppo_loss (loss on a given mu) and bg_actor_head (output layer + loss + output-layer backward in one kernel).
#pragma once
#include <hip/hip_runtime.h>

namespace bg {

constexpr float kHalfLog2Pi = 0.9189385332046727f;

template <int A>
struct ActorLossConsts {
    float logstd[A], old_logstd[A], isig2[A], osig2[A];
    float ent, mean, inv_std, invB, e_clip, bscale;
};

// adv_stats = (sum adv, sum adv^2, count): advantage normalisation (adv - mean) / (std + 1e-8) with torch.std's unbiased estimator (runner.py:145)
template <int A>
__device__ __forceinline__ void actor_loss_consts(ActorLossConsts<A>& c, int B, const float* __restrict__ logstd, const float* __restrict__ old_logstd,
                                                  const double* __restrict__ adv_stats, float e_clip, float bound_coef) {
    const double cnt = adv_stats[2], mean_d = adv_stats[0] / cnt;
    double var_d = (adv_stats[1] - cnt * mean_d * mean_d) / (cnt - 1.0);
    if (var_d < 0.0) var_d = 0.0;
    c.mean = (float)mean_d;
    c.inv_std = 1.0f / ((float)sqrt(var_d) + 1e-8f);
    c.invB = 1.0f / (float)B;
    c.e_clip = e_clip;
    c.bscale = bound_coef * 2.0f * c.invB / (float)A;
    c.ent = 0.f;
    for (int a = 0; a < A; a++) {
        const float sg = expf(logstd[a]), os = expf(old_logstd[a]);
        c.logstd[a] = logstd[a]; c.old_logstd[a] = old_logstd[a];
        c.isig2[a] = 1.0f / (sg * sg); c.osig2[a] = os * os;
        c.ent += 0.5f + kHalfLog2Pi + logstd[a];
    }
}

// One sample.  In: mu m[A], action act[A], old mu om[A], old log-prob, raw advantage.  Out: gmu[A] = dL/dmu, and the sample's contributions
// acc[0..A) = dL/dlogstd, acc[A+1..A+4] = (surrogate, bound penalty, entropy, kl); acc[A] (value error) is the caller's.
template <int A>
__device__ __forceinline__ void actor_loss_row(const ActorLossConsts<A>& c, const float (&m)[A], const float (&act)[A], const float (&om)[A],
                                               float old_logp, float adv, float (&gmu)[A], double (&acc)[A + 5]) {
    float d[A];
    float logp = 0.f, kl = 0.f, bound = 0.f;
    for (int a = 0; a < A; a++) {
        d[a] = act[a] - m[a];
        logp += -0.5f * d[a] * d[a] * c.isig2[a] - c.logstd[a] - kHalfLog2Pi;
        const float dm = m[a] - om[a];
        kl += c.logstd[a] - c.old_logstd[a] + 0.5f * (c.osig2[a] + dm * dm) * c.isig2[a] - 0.5f;
        const float hi = fmaxf(m[a] - 1.0f, 0.f), lo = fminf(m[a] + 1.0f, 0.f);
        bound += hi * hi + lo * lo;
    }
    const float An = (adv - c.mean) * c.inv_std;
    const float ratio = expf(logp - old_logp);
    const float rc = fminf(fmaxf(ratio, 1.0f - c.e_clip), 1.0f + c.e_clip);
    const float s1 = -An * ratio, s2 = -An * rc;
    const float actor = fmaxf(s1, s2);
    // d max(s1,s2)/d logp: through s1 when it wins or ties, through the clamp only inside the clip range
    const bool inside = ratio >= 1.0f - c.e_clip && ratio <= 1.0f + c.e_clip;
    const float dlogp = (inside || s1 > s2) ? -An * ratio * c.invB : 0.f;
    for (int a = 0; a < A; a++) {
        const float hi = fmaxf(m[a] - 1.0f, 0.f), lo = fminf(m[a] + 1.0f, 0.f);
        gmu[a] = dlogp * d[a] * c.isig2[a] + c.bscale * (hi + lo);
        acc[a] += (double)(dlogp * (d[a] * d[a] * c.isig2[a] - 1.0f));
    }
    acc[A + 1] += (double)actor;
    acc[A + 2] += (double)bound;
    acc[A + 3] += (double)c.ent;
    acc[A + 4] += (double)kl;
}

}  // namespace bg
