/* filter_item.h -- one unit of work of the polar filter: a strip of ocean points of one level of one row */
#ifndef UVIC_FILTER_ITEM_H
#define UVIC_FILTER_ITEM_H
typedef struct FilterItem {
  int j, k, is, im; /* row, level, first column, length (columns wrap cyclically past imt-1) */
  int mode;         /* 0: replace by the strip mean (filtr.F:196-203), 1: apply the operator to the deviation from the mean;
                     * velocities (filuv.F, filter type 2): 2: zero (filtr.F:183-188), 3: apply the operator as it is */
  int mat;          /* offset (in doubles) of the compact im x im operator, mode 1 */
  double fnorm, fimr;
  double fx;        /* velocities: -1 south of the equator, +1 north (filuv.F:66-67) */
} FilterItem;
#endif
