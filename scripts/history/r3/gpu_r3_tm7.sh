cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/tmtrace; mkdir -p gpurun_out/tmtrace
for v in "" "--tm_acts"; do
  tag=old; [ -n "$v" ] && tag=new
  timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/tmtrace -o $tag -- python bench.py --steps 2 --warmup 1 --num_steps 50 --batch 32 --dtype bf16 --no_cpu_baseline --no_roofline --no_secondary $v > gpurun_out/tmtrace/run_$tag.log 2>&1
  python scripts/prof_summary.py $(find gpurun_out/tmtrace -name "${tag}_results.db" | head -1) > gpurun_out/tmtrace/${tag}_summary.txt 2>&1
  grep '^{' gpurun_out/tmtrace/run_$tag.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['ms_per_step'])"
done
find gpurun_out/tmtrace -name "*.db" -delete
head -24 gpurun_out/tmtrace/new_summary.txt | cut -c1-150
sed -n '/one denoise step/,$p' gpurun_out/tmtrace/new_summary.txt | cut -c1-170 | head -50
