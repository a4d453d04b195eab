#!/bin/bash
# tools/isa/extract_kernel.sh <listing.s> <substring of the mangled kernel name> : the body of one kernel of an ISA listing on stdout
awk -v k="$2" '$0 ~ "^_Z" && $0 ~ k && $0 ~ /:/ {f=1} f{print} f && /^\.Lfunc_end/{exit}' "$1"
