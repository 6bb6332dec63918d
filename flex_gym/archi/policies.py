from high_speed_quadrupedal_locomotion_by_irrl_amd.policies import (ActorCriticPolicy, CustomLSTMPolicy, LstmPolicy,  # noqa: F401
                                                                     MlpPolicy)
