set -e
K=${1:-/dev/shm/ecseg_keep}
N=${2:-512}
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; nproc; python -c "import os;print(len(os.sched_getaffinity(0)))"
cat /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' '; echo
python tools/time_cli.py --keep $K --n $N --base 16 --batch 32 > gpurun_out/cli_t1.json 2> gpurun_out/cli_t1.err
cat /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' '; echo
python tools/time_cli.py --keep $K --n $N --base 16 --batch 32 --io-threads 8 > gpurun_out/cli_t2.json 2> gpurun_out/cli_t2.err
cat /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' '; echo
rm -rf $K
