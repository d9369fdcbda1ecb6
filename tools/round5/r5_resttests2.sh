#!/bin/bash
# the GPU suite from the convergence file's last test on (after a fix to that test), with timing
cd $GRAFT_REPO_ROOT
( time timeout 3400 python -m pytest tests/ -x -q -m gpu --durations=8 --deselect tests/test_gpu_convergence.py::test_llff_pose_error_curve_matches_the_oracle_loop --deselect tests/test_gpu_convergence.py::test_joint_optimisation_recovers_the_cameras --deselect tests/test_gpu_convergence.py::test_pose_error_curve_matches_the_oracle_loop --deselect tests/test_gpu_convergence.py::test_llff_joint_optimisation_recovers_camera_centres --deselect tests/test_gpu_convergence.py::test_llff_short_schedule_with_the_oracle_render_ends_where_the_hip_path_ends ) > gpurun_out/r5_resttests2.log 2>&1
tail -n 22 gpurun_out/r5_resttests2.log
