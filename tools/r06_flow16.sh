L=oracle/_ref/flow_launcher
for t in 16 16 1; do echo "== libflow_mi355x.so threads $t"; timeout 200 $L oracle/_ref/libflow_mi355x.so --loci 768 --seed 100 --threads $t 2>&1 | tail -1; done
echo "== libflow_mi355x.so threads 16 stream"; timeout 200 $L oracle/_ref/libflow_mi355x.so --loci 768 --seed 100 --threads 16 --stream 2>&1 | tail -1
echo "== batched 16"; timeout 200 $L oracle/_ref/libflow_mi355x_batched.so --loci 768 --seed 100 --threads 16 2>&1 | tail -1
echo "== batched 16 stream"; timeout 200 $L oracle/_ref/libflow_mi355x_batched.so --loci 768 --seed 100 --threads 16 --stream 2>&1 | tail -1
