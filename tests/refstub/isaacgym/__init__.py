"""Test-only stand-in for the closed-source `isaacgym` package.

Purpose: let the reference's *own* env glue (humanoid/envs/base/legged_robot.py,
humanoid/envs/custom/hector_env.py) execute unmodified in this container, with the rigid-body step
underneath it served by oracle/physics.py, so that golden input/output vectors for the observation /
reward / reset / termination arithmetic come from the reference code itself
(tests/golden/make_env_fixtures.py).  It implements only the call surface listed in SURVEY.md 8(b)
"Seam 2".  It is never imported by the product and never travels as anything but test tooling.
"""
