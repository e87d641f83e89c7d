#!/bin/bash
# GPU box: fixed costs of one CLI process (dynamic linking, HIP start-up, exit) next to a 5 GB run.
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=$R/circkit_amd/circkit
printf ">a\nACGTTGCA\n" > /dev/shm/tiny.fasta
t() { s=$(date +%s.%N); "$@" > /dev/null 2>&1; e=$(date +%s.%N); python3 -c "print('%-60s %.3f s' % ('$(echo $@ | cut -c1-60)', $e - $s))"; }
t $C --help
t $C --help
t $C canonicalize /dev/shm/tiny.fasta -o /dev/shm/tiny.out
t $C canonicalize /dev/shm/tiny.fasta -o /dev/shm/tiny.out
LD_DEBUG=statistics $C --help 2>&1 | grep -i "total startup\|relocation" | head -4
