#!/bin/bash
# ablation builds of the member-gradient kernel (csrc/ablate.hpp): build_ab/lib_<name>.so, four at a time
cd "$(dirname "$0")/.."
build() { bash tools/ab_variant.sh "$@" > /dev/null 2>&1 || echo "FAILED $1"; }
build base &
build m_nogst -DIHG_ABL_M_NO_G_STORES &
build m_nodst -DIHG_ABL_M_NO_DOUT_STORE &
build m_nostores -DIHG_ABL_M_NO_G_STORES -DIHG_ABL_M_NO_DOUT_STORE &
wait
build m_nomem -DIHG_ABL_M_NO_MEMBER_LOADS &
build m_nody -DIHG_ABL_M_NO_DY_LOADS &
build m_noloads -DIHG_ABL_M_NO_MEMBER_LOADS -DIHG_ABL_M_NO_DY_LOADS &
build m_nomfma -DIHG_ABL_M_NO_MFMA &
wait
build m_nosplit -DIHG_ABL_M_NO_SPLIT &
build m_nour -DIHG_ABL_M_NO_USER_SUMS &
build m_noprod -DIHG_ABL_M_NO_PRODUCT_RULE -DIHG_ABL_M_NO_USER_SUMS &
build m_nomemory -DIHG_ABL_M_NO_MEMBER_LOADS -DIHG_ABL_M_NO_DY_LOADS -DIHG_ABL_M_NO_G_STORES -DIHG_ABL_M_NO_DOUT_STORE &
wait
build m_svconly -DIHG_ABL_M_NO_MEMBER_LOADS -DIHG_ABL_M_NO_DY_LOADS -DIHG_ABL_M_NO_G_STORES -DIHG_ABL_M_NO_DOUT_STORE -DIHG_ABL_M_NO_MFMA &
build m_mfmaonly -DIHG_ABL_M_NO_MEMBER_LOADS -DIHG_ABL_M_NO_DY_LOADS -DIHG_ABL_M_NO_G_STORES -DIHG_ABL_M_NO_DOUT_STORE -DIHG_ABL_M_NO_SPLIT -DIHG_ABL_M_NO_PRODUCT_RULE -DIHG_ABL_M_NO_USER_SUMS &
wait
ls -la build_ab/*.so | wc -l
