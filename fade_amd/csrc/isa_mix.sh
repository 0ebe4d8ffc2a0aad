#!/bin/bash
# usage: isa_mix.sh <mangled-kernel-prefix> [rows]  — instruction mix of the kernel's innermost (deepest) loop
S=/root/repo/fade_amd/csrc/build/fadehip-hip-amdgcn-amd-amdhsa-gfx950.s
awk -v k="^$1" '$0 ~ k && /:/ && !p {p=1} p {print} p && /s_endpgm/ {exit}' $S > /tmp/kern.s
D=$(grep -o "Inner Loop Header: Depth=[0-9]*" /tmp/kern.s | sort -t= -k2 -n | tail -1)
A=$(grep -n "$D" /tmp/kern.s | tail -1 | cut -d: -f1)
[ -z "$A" ] && { echo "loop not found"; exit 1; }
B=$(awk -v a=$A 'NR > a && /s_cbranch_(scc|vcc)/ {print NR; exit}' /tmp/kern.s)
sed -n "${A},${B}p" /tmp/kern.s > /tmp/loop.s
grep -E "^\s+[vs]_|^\s+ds_|^\s+global_|^\s+buffer_" /tmp/loop.s | awk '{print $1}' | sort | uniq -c | sort -rn | head -${2:-18}
echo "total VALU in loop: $(grep -cE '^\s+v_' /tmp/loop.s)   s_nop: $(grep -c s_nop /tmp/loop.s)"
