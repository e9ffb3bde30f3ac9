// learn_i3.hip -- persistent learner, cartpole (21 actions, 1001-point universes: 10-bit indices, 40 KB of VE tables in LDS), 2 .. 8 lanes per agent
#include "learn_kernel.h"

void frirl_learn_launch_cartpole_lo(int H, const frirl_hip_tables *t, const frirl_hip_rulebases *b, const frirl_hip_agent *ag, const frirl_hip_envs *ev,
                                    const frirl_hip_convergence *cv, const frirl::LearnArgs &la, hipStream_t s)
{
    launch_learn_h<5, 21, FRIRL_HIP_ENV_CARTPOLE, 10, 2, 2, 8>(H, t, b, ag, ev, cv, la, s);
}
