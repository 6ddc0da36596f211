/* filter_item.h -- one unit of work of the polar filter: a strip of ocean points of one level of one row */
#ifndef UVIC_FILTER_ITEM_H
#define UVIC_FILTER_ITEM_H
typedef struct FilterItem {
  int j, k, is, im; /* row, level, first column, length (columns wrap cyclically past imt-1) */
  int mode;         /* 0: replace by the strip mean (filtr.F:196-203), 1: apply the operator */
  int mat;          /* offset (in doubles) of the compact im x im operator, mode 1 */
  double fnorm, fimr;
} FilterItem;
#endif
