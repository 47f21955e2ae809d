#!/bin/bash
# usage: kfunc.sh <asm.s> <mangled-prefix>  -> body of the first function whose label starts with the prefix
awk -v p="^$2" '$0 ~ p && /: *;/ {on=1} on {print} on && /s_endpgm/ {exit}' $1
