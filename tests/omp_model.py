"""The reference's many-agent loop WITH rule-base exchange (frirl_omp_run, reference src/frirl/frirl_agent.c:294-385, running
frirl_sequential_run in chunks, src/frirl/frirl_sequential_run.c:24-165 with runmode FRIRL_OMP) written with the oracle's pieces
(orc_episode, orc_merge_rb, orc_gen_def_states -- each pinned against the genuine reference).  The loop itself is pinned against
the GENUINE frirl_omp_run (tests/golden/omprun_*.jsonl, produced by oracle/ref_merge_harness in `omprun` mode); it is then the
checker of the device's merged training (frirl_hip_batch_train_merged).

What the loop does, as the reference does it (the points the round-2 review found the port deviating from):
  * every agent runs a chunk of at most FRIRL_AGENT_EPCHUNK - 1 = 9 episodes per round; a chunk ends early when the rule base is
    "considered complete" (is_running = 0) -- and such an agent simply runs its NEXT chunk in the next round (frirl_sequential_run
    does not look at is_running on entry, :57-66): only the master's is_running ends the job (:333-336);
  * `epended` is reset on entry of every chunk and set by the CHEAP test alone (same rules / steps / good reward as the previous
    episode, :83-90) -- also when the tolerance check then finds a consequent that moved; it gates the exchange of THAT round only
    (frirl_agent.c:338,352);
  * the "previous episode" of the convergence test is whatever frirl_desc.reward holds (:66-68): it survives the merge, only the
    rule count and the consequent snapshot are taken afresh (after the merge) at the top of the next episode.
"""
import numpy as np

from oracle import binding as ob

EPCHUNK = 10          # FRIRL_AGENT_EPCHUNK


class Agent:
    def __init__(self, env, trig_mode, maxR=0):
        self.fr = ob.Frirl(env, trig_mode=trig_mode, maxR=maxR)
        self.episode_num = 1          # frirl_init
        self.is_running = 1
        self.epended = 0
        self.last_steps = -1          # omp_init: reward.ep_total_steps = reward.ep_total_value = -1
        self.last_reward = -1.0
        self.episodes_run = 0

    def chunk(self, max_episodes, hp):
        """frirl_sequential_run with runmode FRIRL_OMP (construct part)."""
        f = self.fr.five
        self.epended = 0
        epchunk = 1
        while True:
            if not (epchunk < EPCHUNK):
                if not (self.episode_num < max_episodes):
                    self.is_running = 0
                break
            prev_R, prev_reward, prev_steps = f.R, self.last_reward, self.last_steps
            prev_q = np.array(f.rconc[: f.maxR]).copy()
            self.fr.episode()
            self.episodes_run += 1
            self.last_steps, self.last_reward = self.fr.ep_steps, self.fr.ep_reward
            epend = 0
            if prev_R == f.R and prev_steps == self.last_steps and self.last_reward > hp["reward_good_above"] and prev_reward == self.last_reward:
                epend = 1
                self.epended = 1
                if (np.abs(np.array(f.rconc[: f.R]) - prev_q[: f.R]) >= hp["qdiff_final_tolerance"]).any():
                    epend = 0
            if epend == 1:
                self.is_running = 0
                break
            self.episode_num += 1
            epchunk += 1


def run_omp(env, world, max_episodes, trig_mode=0, maxR=0):
    """Returns (agents, rounds, pended_prints): agents[0] holds the master's final rule base."""
    ag = [Agent(env, trig_mode, maxR) for _ in range(world)]
    ns = ag[0].fr.nstates
    for i in range(1, world):                       # omp_init -> gen_def_states on the initial rule list
        ag[i].fr.set_start_state(ag[0].fr.five.gen_def_states(i, world, ns))
    hp = ag[0].fr.hparams
    rounds = pended = 0
    while True:
        for a in ag:
            a.chunk(max_episodes, hp)
        if ag[0].is_running == 0:
            break
        m = ag[0].fr.five
        if ag[0].epended == 0:
            mr, mc = np.array(m.rant[: m.R]), np.array(m.rconc[: m.R])
            for i in range(1, world):
                ag[i].fr.five.merge_rb(ag[i].fr.agent(), mr, mc)
        else:
            pended += 1
        for i in range(1, world):
            if ag[i].epended == 0:
                f = ag[i].fr.five
                m.merge_rb(ag[0].fr.agent(), np.array(f.rant[: f.R]), np.array(f.rconc[: f.R]))
            else:
                pended += 1
        rounds += 1
    return ag, rounds, pended
