"""Per-episode recorder of env 0: the reference's `Logger`
(tasks/control/logger.py:19-46) + `FpvBase.record` (fpv_asymmetry.py:655-696), fed from the state blob.

Same behaviour as the reference's logger: `store_buffer(**arrays)` appends one row per key, `dump_buffer()` writes
<key><index>.npy and .csv into `output_dir` for at most the first 5 episodes, `reset_buffer()` clears."""
import os

import numpy as np

from . import _lib

_F = dict(POS=0, QUAT=3, LINVEL=7, ANGVEL=10, TGT_POS=13, TGT_QUAT=16, RPY_CONT=23, BAT_V=35, OMEGA=36, ACT=40, ACT_OLD=44, CMD=48)


class EpisodeRecorder:
    MAX_DUMPS = 5  # logger.py:36

    def __init__(self, output_dir):
        self.output_dir = output_dir
        self.episode_dict = {}
        self.dump_index = 0

    def store_buffer(self, **kwargs):
        for k, v in kwargs.items():
            self.episode_dict.setdefault(k, []).append(np.asarray(v))

    def dump_buffer(self):
        if self.dump_index >= self.MAX_DUMPS or not self.episode_dict:
            return False
        os.makedirs(self.output_dir, exist_ok=True)
        for k, v in self.episode_dict.items():
            arr = np.array(v)
            np.save(os.path.join(self.output_dir, f"{k}{self.dump_index}"), arr)
            np.savetxt(os.path.join(self.output_dir, f"{k}{self.dump_index}.csv"), arr.reshape(len(arr), -1), delimiter=",")
        self.dump_index += 1
        return True

    def reset_buffer(self):
        self.episode_dict = {}

    def record(self, env, env_index=0):
        """one row of env `env_index` at the RL rate, with the reference's key names (fpv_asymmetry.py:657-696) for
        everything the kernel keeps; call after env.step().  Starts a new file set when the env was just reset."""
        if not env.tracks_rpy(env_index):
            raise _lib.TacoError("EpisodeRecorder.record: copter_rpy_continuous of this env is not kept up to date -- the step kernel maintains it for "
                                 "flip envs only unless the env was created with cfg['record_flag'] = True (as the reference's test mode does, "
                                 "fpv_asymmetry.py:113-117, train_fpv_asymmetry_ppo.py:349-352)")
        blob = env.get_state()[:_lib.NUM_FIELDS, env_index].cpu().numpy()
        progress = int(blob[65:66].view(np.int32)[0])
        if progress == 1 and self.episode_dict:     # fpv_asymmetry.py:514-517: dump + reset when env 0 is reset
            self.dump_buffer()
            self.reset_buffer()
        self.store_buffer(
            copter_pos=blob[_F["POS"]:_F["POS"] + 3], copter_quat=blob[_F["QUAT"]:_F["QUAT"] + 4],
            copter_linvel=blob[_F["LINVEL"]:_F["LINVEL"] + 3], copter_angvel=blob[_F["ANGVEL"]:_F["ANGVEL"] + 3],
            copter_rpy_continuous=blob[_F["RPY_CONT"]:_F["RPY_CONT"] + 3],
            target_pos=blob[_F["TGT_POS"]:_F["TGT_POS"] + 3], target_quat=blob[_F["TGT_QUAT"]:_F["TGT_QUAT"] + 4],
            battery_voltage=blob[_F["BAT_V"]:_F["BAT_V"] + 1], rotor_speed=blob[_F["OMEGA"]:_F["OMEGA"] + 4],
            command=blob[_F["CMD"]:_F["CMD"] + 2], observations=env.obs_buf[env_index, -1].cpu().numpy(),
            actions=blob[_F["ACT"]:_F["ACT"] + 4], actions_old=blob[_F["ACT_OLD"]:_F["ACT_OLD"] + 4],
            reward=env.rew_buf[env_index:env_index + 1].cpu().numpy(), done=env.reset_buf[env_index:env_index + 1].cpu().numpy())
