#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
bash scripts/variants.sh asm512 asm1024
bash scripts/variants.sh asm512 asm1024
