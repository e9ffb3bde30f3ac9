// learn_i1.hip -- kernel instantiations of the persistent learner (learn_kernel.h), split over files for a parallel build.
#include "learn_kernel.h"

void frirl_learn_launch_acrobot_lo(int H, const frirl_hip_tables *t, const frirl_hip_rulebases *b, const frirl_hip_agent *ag, const frirl_hip_envs *ev,
                                   const frirl_hip_convergence *cv, const frirl::LearnArgs &la, hipStream_t s)
{
    launch_learn_h<5, 3, FRIRL_HIP_ENV_ACROBOT, 6, 2, 1, 8>(H, t, b, ag, ev, cv, la, s);
}
