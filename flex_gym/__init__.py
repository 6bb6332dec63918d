"""Import-path compatibility with the reference package `flex_gym` (FlexibleRobotRaisimGym/flex_gym):
thin re-exports of high_speed_quadrupedal_locomotion_by_irrl_amd so that the reference's import lines
(run_bp_v5.py:8-13) work unchanged."""
