"""Pieces the shared-policy and the separated-policy SMAC runners have in common: the team-level masks derived from
``dones`` / ``infos``, the incremental win rate of the training envs and the bookkeeping of the eval loop.
Semantics are the reference's (runner/shared/smac_runner.py:60-84, :133-149, :196-221; the separated runner
repeats them)."""
import numpy as np

SMAC_ENV_NAMES = ("StarCraft2", "SMACv2", "SMAC", "StarCraft2v2")


def team_masks(dones, infos, num_agents):
    """dones [N, A] bool, infos [N][A] dicts -> (team_done [N] bool, masks, active_masks, bad_masks), each
    [N, A, 1] float32:
      masks         0 for every agent of a team whose episode ended (all agents done), else 1;
      active_masks  0 for an agent that died while its team fights on; a finished team is reset, so all 1 there;
      bad_masks     0 where the env reports ``bad_transition`` (episode cut by the time limit), else 1."""
    dones = np.asarray(dones, dtype=bool)
    team_done = dones.all(axis=1)
    shape = (dones.shape[0], num_agents, 1)
    masks = np.broadcast_to(~team_done[:, None, None], shape).astype(np.float32)
    dead = dones & ~team_done[:, None]
    active_masks = (~dead)[..., None].astype(np.float32)
    bad_masks = np.array([[[0.0] if info[a]['bad_transition'] else [1.0] for a in range(num_agents)]
                          for info in infos], dtype=np.float32)
    return team_done, masks, active_masks, bad_masks


class BattleLog(object):
    """Win rate over the battles fought since the previous report (the envs count battles cumulatively)."""

    def __init__(self, n_threads):
        self.games = np.zeros(n_threads, dtype=np.float32)
        self.wins = np.zeros(n_threads, dtype=np.float32)

    def incremental_win_rate(self, infos):
        wins, games, new_wins, new_games = [], [], [], []
        for i, info in enumerate(infos):
            first = info[0]
            if 'battles_won' in first:
                wins.append(first['battles_won'])
                new_wins.append(first['battles_won'] - self.wins[i])
            if 'battles_game' in first:
                games.append(first['battles_game'])
                new_games.append(first['battles_game'] - self.games[i])
        rate = np.sum(new_wins) / np.sum(new_games) if np.sum(new_games) > 0 else 0.0
        self.games, self.wins = games, wins          # like the reference: the lists, as reported
        return rate


def progress_line(all_args, algorithm_name, experiment_name, episode, episodes, steps_done, steps_total, elapsed):
    return "\n Map {} Algo {} Exp {} updates {}/{} episodes, total num timesteps {}/{}, FPS {}.\n".format(
        getattr(all_args, "map_name", "?"), algorithm_name, experiment_name, episode, episodes, steps_done,
        steps_total, int(steps_done / elapsed))


def mask_entries(active_masks):
    n = 1
    for d in active_masks.shape:
        n *= int(d)
    return n
