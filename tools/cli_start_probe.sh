#!/bin/bash
# GPU box: where do the 0.3 s of `circkit` outside main()'s pipeline go?  Wall time of the binary doing nothing (--version: the
# dynamic loader has mapped libcirckit_hip.so and the HIP runtime behind it, no HIP call is made), of an empty input (HIP start-up,
# context, pinned buffers, exit) and the loader's own statistics.
R=${GRAFT_REPO_ROOT:-$(pwd)}
t() { s=$(date +%s.%N); "$@" > /dev/null 2>&1; e=$(date +%s.%N); python3 -c "print('%-60s %.3f s' % ('$*'[:60], $e - $s))"; }
for i in 1 2 3; do t $R/circkit_amd/circkit --version; done
printf ">r\nACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT\n" > /dev/shm/one.fasta
for i in 1 2 3; do t $R/circkit_amd/circkit canonicalize /dev/shm/one.fasta -o /dev/null; done
CIRCKIT_CLI_TIMING=2 $R/circkit_amd/circkit canonicalize /dev/shm/one.fasta -o /dev/null 2>&1 | tail -12
LD_DEBUG=statistics $R/circkit_amd/circkit --version 2>&1 | grep -i "total startup\|relocation\|load" | head
ldd $R/circkit_amd/circkit | wc -l
rm -f /dev/shm/one.fasta
