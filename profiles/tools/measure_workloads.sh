set -o pipefail
o=gpurun_out/meas; mkdir -p $o
b() { name=$1; shift; timeout -k 10 250 python bench.py "$@" --no-cpu-baseline > $o/$name.json 2> $o/$name.err || { echo FAIL $name; tail -3 $o/$name.err; exit 1; }; echo $name done; }
b nsvae_kl_bf16 --workload nsvae_kl --precision bf16x3 --batch 32 --steps 10 --warmup 3
IDV_CONCURRENT=1 b nsvae_kl_bf16_conc --workload nsvae_kl --precision bf16x3 --batch 32 --steps 10 --warmup 3
b nsvae_kl_f32 --workload nsvae_kl --batch 32 --steps 6 --warmup 2
b twophase_bf16 --workload twophase --precision bf16x3 --batch 32 --steps 10 --warmup 3
b twophase_f32 --workload twophase --batch 32 --steps 6 --warmup 2
b cvae_bf16 --workload cvae_elbo --precision bf16x3 --steps 8 --warmup 2
b cvae_f32 --workload cvae_elbo --steps 4 --warmup 2
b nsvae_train_f32 --workload nsvae_train --steps 4 --warmup 2
b nsvae_train_bf16 --workload nsvae_train --precision bf16x3 --steps 6 --warmup 2
b twophase_train_f32 --workload twophase_train --steps 4 --warmup 2
b twophase_train_bf16 --workload twophase_train --precision bf16x3 --steps 6 --warmup 2
b cvae_train_f32 --workload cvae_train --steps 4 --warmup 2
b cvae_train_bf16 --workload cvae_train --precision bf16x3 --steps 4 --warmup 2
