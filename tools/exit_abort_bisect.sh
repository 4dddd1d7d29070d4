T=tests/test_gpu_world.py
A=$T::test_world_of_one_equals_single_context
B=$T::test_comm_init_rank_path_equals_single_context
Cc=$T::test_world_refuses_devices_it_does_not_have
run() { name=$1; shift; timeout -k 10 300 python -X faulthandler -m pytest -q -p no:cacheprovider "$@" > gpurun_out/bis_$name.log 2>&1; echo "$name rc=$?"; tail -n 25 gpurun_out/bis_$name.log | grep -E "passed|failed|free|File|Abort" ; }
run abc_gpu -m gpu $A $B $Cc
run abc_nom $A $B $Cc
run ac_nom $A $Cc
run bc_nom $B $Cc
run c_nom $Cc
run whole_gpu -m gpu $T
which gdb || true
