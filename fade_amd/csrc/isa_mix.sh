#!/bin/bash
# usage: isa_mix.sh <mangled-kernel-prefix>   — instruction mix of the kernel's innermost loop
S=/root/repo/fade_amd/csrc/build/fadehip-hip-amdgcn-amd-amdhsa-gfx950.s
awk -v k="^$1" '$0 ~ k":" {p=1} p {print} p && /s_endpgm/ {exit}' $S > /tmp/kern.s
A=$(grep -n "s_cbranch_scc1" /tmp/kern.s | tail -1 | cut -d: -f1)
B=$(grep -n "s_cbranch_scc0" /tmp/kern.s | tail -1 | cut -d: -f1)
[ -z "$A" ] || [ -z "$B" ] && { echo "loop not found ($A,$B)"; exit 1; }
sed -n "${A},${B}p" /tmp/kern.s > /tmp/loop.s
grep -E "^\s+[vs]_|^\s+ds_|^\s+global_|^\s+buffer_" /tmp/loop.s | awk '{print $1}' | sort | uniq -c | sort -rn | head -${2:-18}
echo "total VALU in loop: $(grep -cE '^\s+v_' /tmp/loop.s)"
