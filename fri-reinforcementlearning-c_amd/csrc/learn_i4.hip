// learn_i4.hip -- persistent learner, cartpole, 16 .. 64 lanes per agent
#include "learn_kernel.h"

void frirl_learn_launch_cartpole_hi(int H, const frirl_hip_tables *t, const frirl_hip_rulebases *b, const frirl_hip_agent *ag, const frirl_hip_envs *ev,
                                    const frirl_hip_convergence *cv, const frirl::LearnArgs &la, hipStream_t s)
{
    launch_learn_h<5, 21, FRIRL_HIP_ENV_CARTPOLE, 10, 2, 16, 64>(H, t, b, ag, ev, cv, la, s);
}
