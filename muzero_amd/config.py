"""Hyper-parameters of the search and training paths, with the reference's attribute names
(model.py:13-81) so that a reference ``MuZeroConfig`` object can be passed
wherever this one is accepted."""
from typing import List, Tuple


class MuZeroConfig:
    def __init__(self):
        # environment
        self.env_name: str = "ALE/Pong-v5"
        self.observation_shape: Tuple[int, int, int] = (96, 96, 12)  # [0] is read as input channels (model.py:181)
        self.action_space_size: int = 6
        self.num_players: int = 1
        self.support_size: int = 300
        # network
        self.num_blocks: int = 2
        self.num_channels: int = 256
        self.reduced_channels_reward: int = 256
        self.reduced_channels_value: int = 256
        self.reduced_channels_policy: int = 256
        self.fc_reward_layers: List[int] = [256]
        self.fc_value_layers: List[int] = [256]
        self.fc_policy_layers: List[int] = [256]
        self.downsample_steps: List[Tuple[int, int]] = [(128, 2), (256, 3)]
        # self-play and training
        self.training_steps: int = int(1000e3)
        self.checkpoint_interval: int = 1000
        self.batch_size: int = 2
        self.num_unroll_steps: int = 5
        self.td_steps: int = 10
        self.num_actors: int = 2
        self.max_moves: int = 27000
        self.num_simulations: int = 5
        # replay buffer
        self.replay_buffer_size: int = int(1e6)
        self.minimum_games_in_buffer: int = 2
        self.priority_alpha: float = 1.0
        self.priority_beta: float = 1.0
        # search
        self.pb_c_base: float = 19652
        self.pb_c_init: float = 1.25
        self.root_dirichlet_alpha: float = 0.3
        self.root_exploration_fraction: float = 0.25
        self.discount: float = 0.997
        # optimisation
        self.weight_decay: float = 1e-4
        self.momentum: float = 0.9
        self.lr_init: float = 0.05
        self.lr_decay_rate: float = 0.1
        self.lr_decay_steps: int = int(350e3)
        self.value_loss_weight: float = 0.25
        self.reward_loss_weight: float = 1.0
        self.policy_loss_weight: float = 1.0
        self.max_grad_norm: float = 10.0

    def visit_softmax_temperature_fn(self, trained_steps: int) -> float:
        """Temperature schedule of model.py:74-81."""
        if trained_steps < 500e3:
            return 1.0
        if trained_steps < 750e3:
            return 0.5
        return 0.25
